#!/bin/bash
# Static count of float64 VALU instructions per kernel of one .hip source (the OLS kernels' unit
# loops are straight-line code, so this is the per-unit count plus a small prologue).
#   scripts/count_dp.sh fir_ols32.hip [extra hipcc flags...]
cd "$(dirname "$0")/../pipe_amd/csrc"
SRC=${1:-fir_ols32.hip}; shift
TMP=$(mktemp -d)
/opt/rocm/bin/hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -I../../include -I. "$@" -S --cuda-device-only "$SRC" -o "$TMP/dev.s" || exit 1
awk '
/^_Z[A-Za-z0-9_]+:/ { name=$1; next }
/^\.Lfunc_end/ { name="" }
name != "" {
  if ($1 ~ /^v_(fma|fmac|mul|add|max|min)_f64/) dp[name]++;
  if ($1 ~ /^v_cvt_f64|^v_cvt_f32_f64/) cv[name]++;
  if ($1 ~ /^ds_/) ds[name]++;
  if ($1 ~ /^v_/) v[name]++;
  if ($1 ~ /^scratch_/) sc[name]++;
  if ($1 ~ /^[a-z]/) tot[name]++;
}
END { for (n in tot) printf "%-60s dp %5d cvt %4d ds %4d valu %5d scratch %4d all %6d\n", substr(n,1,60), dp[n], cv[n], ds[n], v[n], sc[n], tot[n] }' "$TMP/dev.s" | sort
rm -rf "$TMP"
