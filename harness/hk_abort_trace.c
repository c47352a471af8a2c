/* hk_abort_trace.c -- diagnostic helper of the test / bench harness (harness/_build/libhk_abort_trace.so; NOT part of the product).
 *
 * A GPU process that dies of SIGABRT usually dies on a native thread of the HSA runtime (memory fault, queue error,
 * hardware exception: the runtime prints one line to fd 2 and calls abort()).  Under pytest that line is lost when fd 2
 * is captured, and Python's faulthandler can only show the Python frames of OTHER threads.  hk_abort_trace_install()
 * puts a SIGABRT / SIGSEGV / SIGBUS handler in front of whatever is installed (faulthandler's) that writes to `out_fd`
 * (a duplicate of the real stderr taken by the caller) and, when given, appends to the file `path`:
 *   - si_code / si_pid (who sent the signal: SI_TKILL from this process = abort() / raise()), pid, tid, thread name;
 *   - the native backtrace of the receiving thread (backtrace_symbols_fd: module + offset);
 *   - the tail of fd 2 when fd 2 is a regular file (pytest's capture file: the runtime's own last words);
 * then restores the previous disposition and re-raises on the same thread, so faulthandler and the core dump follow.
 *   gcc -O1 -g -shared -fPIC -o libhk_abort_trace.so hk_abort_trace.c                                               */
#define _GNU_SOURCE
#include <execinfo.h>
#include <fcntl.h>
#include <signal.h>
#include <stdio.h>
#include <string.h>
#include <sys/stat.h>
#include <sys/syscall.h>
#include <unistd.h>

static int g_fd = 2;
static char g_path[512];
static struct sigaction g_prev[65];
static volatile int g_installed;

static void emit(int fd, const char* s, size_t n) {
    while (n > 0) {
        ssize_t w = write(fd, s, n);
        if (w <= 0) return;
        s += w, n -= (size_t)w;
    }
}

static void report(int fd, int sig, siginfo_t* si) {
    char line[256], name[32] = "?";
    int c = open("/proc/thread-self/comm", O_RDONLY);
    if (c >= 0) {
        ssize_t n = read(c, name, sizeof name - 1);
        if (n > 0) name[n - 1] = 0;
        close(c);
    }
    int n = snprintf(line, sizeof line,
                     "\n[hk_abort_trace] signal %d si_code %d si_pid %d (self %d) tid %ld thread '%s'%s\n", sig,
                     si ? si->si_code : 0, si ? (int)si->si_pid : 0, (int)getpid(), (long)syscall(SYS_gettid), name,
                     (si && si->si_code == SI_TKILL && si->si_pid == getpid()) ? " -- raised by this process (abort/raise)"
                                                                                : "");
    if (n > 0) emit(fd, line, (size_t)n);
    void* frames[64];
    int depth = backtrace(frames, 64);
    backtrace_symbols_fd(frames, depth, fd);
    /* what the process itself last wrote to a captured stderr */
    struct stat st;
    if (fd != 2 && fstat(2, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) {
        char buf[4096];
        off_t from = st.st_size > (off_t)sizeof buf ? st.st_size - (off_t)sizeof buf : 0;
        ssize_t got = pread(2, buf, sizeof buf, from);
        if (got > 0) {
            static const char head[] = "[hk_abort_trace] tail of the captured stderr (fd 2):\n";
            emit(fd, head, sizeof head - 1);
            emit(fd, buf, (size_t)got);
            emit(fd, "\n", 1);
        }
    }
    static const char tail[] = "[hk_abort_trace] end\n";
    emit(fd, tail, sizeof tail - 1);
}

static void on_fatal(int sig, siginfo_t* si, void* uc) {
    (void)uc;
    report(g_fd, sig, si);
    if (g_path[0]) {
        int f = open(g_path, O_WRONLY | O_CREAT | O_APPEND, 0644);
        if (f >= 0) {
            report(f, sig, si);
            close(f);
        }
    }
    /* hand over to whoever was there before (faulthandler, then the default action) on this same thread */
    if (sig >= 0 && sig < 65) sigaction(sig, &g_prev[sig], NULL);
    syscall(SYS_tgkill, getpid(), syscall(SYS_gettid), sig);
}

/* out_fd < 0: fd 2.  path NULL or "": no file.  Returns 0, or -1 if a handler could not be installed.  Idempotent. */
int hk_abort_trace_install(int out_fd, const char* path) {
    if (g_installed) return 0;
    void* warm[4];
    (void)backtrace(warm, 4); /* loads the unwinder now, not inside the handler */
    g_fd = out_fd >= 0 ? out_fd : 2;
    g_path[0] = 0;
    if (path && *path) {
        strncpy(g_path, path, sizeof g_path - 1);
        g_path[sizeof g_path - 1] = 0;
    }
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = on_fatal;
    sa.sa_flags = SA_SIGINFO | SA_NODEFER | SA_ONSTACK;
    sigemptyset(&sa.sa_mask);
    int rc = 0;
    const int sigs[3] = {SIGABRT, SIGSEGV, SIGBUS};
    for (int i = 0; i < 3; ++i)
        if (sigaction(sigs[i], &sa, &g_prev[sigs[i]]) != 0) rc = -1;
    g_installed = 1;
    return rc;
}
