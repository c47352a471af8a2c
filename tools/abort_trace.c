/* LD_PRELOAD diagnostic for a process that dies of SIGABRT / SIGSEGV without a message: says which thread raised the signal and
 * from where.  It (1) interposes abort / raise / kill / pthread_kill / tgkill-by-syscall callers' usual entry points and prints
 * the caller's native backtrace before passing SIGABRT on, (2) reports every change of the SIGABRT disposition, (3) installs a
 * handler of its own that prints the backtrace of the receiving thread and then lets the default action happen.
 *   gcc -O1 -g -shared -fPIC -o _ab/abort_trace.so tools/abort_trace.c -ldl
 *   LIBC_FATAL_STDERR_=1 LD_PRELOAD=$PWD/_ab/abort_trace.so python -m pytest ...                                          */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <execinfo.h>
#include <pthread.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/syscall.h>
#include <unistd.h>

static void say(const char* what, int sig, long extra) {
    char line[200], name[32] = "?";
    int fd = (int)syscall(SYS_open, "/proc/thread-self/comm", 0);
    if (fd >= 0) {
        long n = read(fd, name, sizeof name - 1);
        if (n > 0) name[n - 1] = 0;
        close(fd);
    }
    int n = snprintf(line, sizeof line, "\n[abort_trace] %s sig %d (%ld) pid %d tid %ld thread '%s'\n", what, sig, extra,
                     (int)getpid(), (long)syscall(SYS_gettid), name);
    if (n > 0) (void)!write(2, line, (size_t)n);
    void* frames[96];
    int depth = backtrace(frames, 96);
    backtrace_symbols_fd(frames, depth, 2);
}

static void on_fatal(int sig, siginfo_t* si, void* uc) {
    (void)uc;
    say("handler: received", sig, si ? (long)si->si_pid * 1000 + (si->si_code & 0xff) : 0);
    struct sigaction dfl;
    memset(&dfl, 0, sizeof dfl);
    dfl.sa_handler = SIG_DFL;
    typedef int (*sigaction_t)(int, const struct sigaction*, struct sigaction*);
    sigaction_t real = (sigaction_t)dlsym(RTLD_NEXT, "sigaction");
    real(sig, &dfl, NULL);
    syscall(SYS_tgkill, getpid(), syscall(SYS_gettid), sig);
}

void abort(void) {
    say("abort() called", SIGABRT, 0);
    void (*real)(void) = (void (*)(void))dlsym(RTLD_NEXT, "abort");
    real();
    _exit(134);
}

int raise(int sig) {
    if (sig == SIGABRT) say("raise() called", sig, 0);
    int (*real)(int) = (int (*)(int))dlsym(RTLD_NEXT, "raise");
    return real(sig);
}

int kill(pid_t pid, int sig) {
    if (sig == SIGABRT) say("kill() called", sig, (long)pid);
    int (*real)(pid_t, int) = (int (*)(pid_t, int))dlsym(RTLD_NEXT, "kill");
    return real(pid, sig);
}

int pthread_kill(pthread_t t, int sig) {
    if (sig == SIGABRT) say("pthread_kill() called", sig, 0);
    int (*real)(pthread_t, int) = (int (*)(pthread_t, int))dlsym(RTLD_NEXT, "pthread_kill");
    return real(t, sig);
}

int sigaction(int sig, const struct sigaction* act, struct sigaction* old) {
    if (sig == SIGABRT && act) say("sigaction(SIGABRT) set to", sig, (long)(size_t)act->sa_handler);
    typedef int (*sigaction_t)(int, const struct sigaction*, struct sigaction*);
    sigaction_t real = (sigaction_t)dlsym(RTLD_NEXT, "sigaction");
    return real(sig, act, old);
}

sighandler_t signal(int sig, sighandler_t h) {
    if (sig == SIGABRT) say("signal(SIGABRT) set to", sig, (long)(size_t)h);
    sighandler_t (*real)(int, sighandler_t) = (sighandler_t(*)(int, sighandler_t))dlsym(RTLD_NEXT, "signal");
    return real(sig, h);
}

__attribute__((constructor)) static void install(void) {
    void* warm[4];
    (void)backtrace(warm, 4); /* loads libgcc's unwinder now, not inside a handler */
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = on_fatal;
    sa.sa_flags = SA_SIGINFO | SA_NODEFER;
    sigemptyset(&sa.sa_mask);
    typedef int (*sigaction_t)(int, const struct sigaction*, struct sigaction*);
    sigaction_t real = (sigaction_t)dlsym(RTLD_NEXT, "sigaction");
    real(SIGABRT, &sa, NULL);
    real(SIGSEGV, &sa, NULL);
    real(SIGBUS, &sa, NULL);
}
