#!/bin/bash
# gfx950 ISA of one built object: tools/disasm.sh conv_mfma > /tmp/conv_mfma.s
set -e
L=/opt/rocm/lib/llvm/bin; T=$(mktemp -d)
$L/llvm-objcopy --dump-section .hip_fatbin=$T/fat.bin mcquic_amd/_obj/$1.o $T/copy.o
$L/clang-offload-bundler --unbundle --type=o --input=$T/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/dev.co
$L/llvm-objdump -d --mcpu=gfx950 $T/dev.co | c++filt
rm -rf $T
