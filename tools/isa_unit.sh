#!/bin/bash
# usage: tools/isa_unit.sh k_verify_var.hip <kernel-substring> [extra hipcc flags...]   -> static ISA loop summary (tools/isa_loops.py)
set -e
unit=$1; kern=$2; shift 2
mkdir -p /tmp/isa_unit && cd /tmp/isa_unit
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden "$@" -c /root/repo/bp_pp_amd/csrc/$unit -o u.o -save-temps 2>/dev/null
python3 /root/repo/tools/isa_loops.py ${unit%.hip}-hip-amdgcn-amd-amdhsa-gfx950.s "$kern"
