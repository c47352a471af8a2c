#!/bin/bash
# dis.sh NAME KERNEL_MANGLED_PREFIX -> head_NAME.s
cd /root/repo/_ab
rm -f hk_kernels_$1.o.0.*
/opt/rocm/lib/llvm/bin/llvm-objdump --offloading hk_kernels_$1.o > /dev/null
/opt/rocm/lib/llvm/bin/llvm-objdump -d hk_kernels_$1.o.0.hipv4-amdgcn-amd-amdhsa--gfx950 > $1.s
k=${2:-_ZN2hk16fit_apply_kernelILi2ELb1ELi2ELb1ELi1ELb1ELi1}
awk -v k="$k" '/^[0-9a-f]+ </{p = index($0, k) > 0} p' $1.s > head_$1.s
wc -l head_$1.s
