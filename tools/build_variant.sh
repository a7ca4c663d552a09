#!/bin/bash
# tools/build_variant.sh <name> <source.hip> <extra flags...>: clap_amd/lib_ab/<name>/libclapgpu.so = the shipped objects
# with <source> rebuilt under the extra flags (A/B builds for tools/kernel_ab.sh)
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; src=$2; shift 2
d=$R/clap_amd/lib_ab/$name
mkdir -p "$d" && cp "$R"/clap_amd/lib/*.o "$d"/ && rm -f "$d/${src%.hip}.o" "$d/libclapgpu.so"
make -s -C "$R/clap_amd/csrc" OUT=../lib_ab/$name EXTRA="$*" ../lib_ab/$name/libclapgpu.so 2>&1 | grep -i "error" ; ls -la "$d/libclapgpu.so" | awk '{print $5, $9}'
