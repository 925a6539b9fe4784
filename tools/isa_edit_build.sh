#!/bin/bash
# usage: tools/isa_edit_build.sh <libdir under ntsc-crt_amd/> <unit> [extra hipcc flags...]
# EXPERIMENT TOOLING (not part of the product build): builds ntsc-crt_amd/<libdir>/libcrthip.so like the Makefile does, except that the
# translation unit csrc/<unit>.hip goes through its assembly text: device code to .s, tools/strip_asm_nops.py over it (drops the one
# wait state the compiler puts behind every inline-asm block whose result the next instruction reads -- it has to assume the block
# wrote a partial register (dst_sel forwarding, gfx940+); blocks that do hold an SDWA / op_sel destination keep theirs), then
# (EDIT=pad: tools/pad_dependent_valu.py instead -- a wait state between every two adjacent dependent vector instructions), then
# (EDIT=none: no edit -- for device-only compiler options, given in $DEVFLAGS, that the host pass does not take), then
# assembler, device link, bundle and the host half of the unit with that bundle embedded -- the same steps `hipcc -###` prints.
# The other objects are taken from ntsc-crt_amd/lib/ (build that first), or from ntsc-crt_amd/$BASE/ to stack several edited units.
set -e
root="$(cd "$(dirname "$0")/.." && pwd)"
libdir=$1; unit=$2; shift 2
LL=/opt/rocm/lib/llvm/bin
src=$root/ntsc-crt_amd/csrc/$unit.hip
FL="-O3 -std=c++17 -fwrapv -fPIC -I$root/include -I$root/ntsc-crt_amd/csrc $*"
out=$root/ntsc-crt_amd/$libdir
tmp=$(mktemp -d /tmp/isaedit_XXXX)
mkdir -p $out
for o in crt_encode crt_noise crt_sync crt_decode crt_decode2 crt_decode3 crt_decode4 crt_host crt_setup; do cp $root/ntsc-crt_amd/${BASE:-lib}/$o.o $out/ 2>/dev/null || true; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 $FL $DEVFLAGS --cuda-device-only -S $src -o $tmp/dev.s 2>/dev/null
if [ "${EDIT:-strip}" = pad ]; then python3 $root/tools/pad_dependent_valu.py $tmp/dev.s $tmp/dev2.s; elif [ "${EDIT:-strip}" = none ]; then cp $tmp/dev.s $tmp/dev2.s; else python3 $root/tools/strip_asm_nops.py $tmp/dev.s $tmp/dev2.s; fi
$LL/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $tmp/dev2.s -o $tmp/dev.o
$LL/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared $tmp/dev.o -o $tmp/dev.out
$LL/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$tmp/dev.out -output=$tmp/dev.hipfb
/opt/rocm/bin/hipcc --offload-arch=gfx950 $FL --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $tmp/dev.hipfb -c $src -o $out/$unit.o 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libcrthip.so $out/crt_encode.o $out/crt_noise.o $out/crt_sync.o $out/crt_decode.o $out/crt_decode2.o $out/crt_decode3.o $out/crt_decode4.o $out/crt_host.o $out/crt_setup.o
rm -rf $tmp
ls -la $out/libcrthip.so
