#!/bin/bash
# usage: tools/kernel_regs.sh <out.txt>   -- VGPR / spill / scratch / LDS of every kernel of libhpgmg_hip.so (device assembly of each .hip file)
out=${1:-/tmp/kernel_regs.txt}; tmp=$(mktemp -d); : > $out
for f in ${ONLY:-hpgmg_amd/csrc/kernels/*.hip}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -Iinclude -Ihpgmg_amd/csrc/kernels $EXTRA -S --cuda-device-only -o $tmp/x.s $f 2>/dev/null
  awk '/\.name:/{name=$2} /\.group_segment_fixed_size:/{l=$2} /\.private_segment_fixed_size:/{p=$2} /\.sgpr_spill_count:/{ss=$2} /\.vgpr_count:/{v=$2} /\.vgpr_spill_count:/{print name, "vgpr="v, "vspill="$2, "sspill="ss, "scratch="p, "lds="l}' $tmp/x.s >> $out
done
rm -rf $tmp; sort -o $out $out; wc -l $out
