#!/bin/bash
# usage: tools/isa.sh <file.hip> <mangled-kernel-substring>  -> resource usage + /tmp/isa_<sub>.s
set -e
cd /root/repo/deeprob-kit_amd/csrc
f=$1; sub=$2
base=$(basename $f .hip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../include -c $f -o /tmp/$base.o -save-temps=obj -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A9 "Function Name: .*$sub" | grep -E "Function Name|SGPRs|VGPRs|AGPRs|Scratch|Occupancy" | head -14
name=$(grep -o "^_Z[A-Za-z0-9_]*$sub[A-Za-z0-9_]*:" /tmp/$base-hip-amdgcn-amd-amdhsa-gfx950.s | head -1 | tr -d ':')
awk "/^$name:/,/s_endpgm/" /tmp/$base-hip-amdgcn-amd-amdhsa-gfx950.s > /tmp/isa_$sub.s
wc -l /tmp/isa_$sub.s
