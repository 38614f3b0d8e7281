#!/bin/bash
# Static VALU instruction histograms of the texture kernels' gfx950 code (no GPU needed): profiles/<round>_glcm_*_valu_hist.txt,
# then profiles/valu_mix.py <round> prices them with the measured issue costs -> profiles/<round>_valu_issue.json.
# The quad kernel is compiled with its per-window finish loop unrolled (RSSEG_GLCM_COUNT_UNROLL) so that static = executed.
# Usage: profiles/valu_hist.sh r04
set -e
ROUND=${1:-r04}
HERE=$(cd "$(dirname "$0")" && pwd)
SRC=$HERE/../rs-image-segmentation_amd/csrc
TMP=$(mktemp -d)
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -S --cuda-device-only"
(cd $SRC && /opt/rocm/bin/hipcc $FLAGS -o $TMP/k4.s k4_glcm.hip 2>/dev/null && /opt/rocm/bin/hipcc $FLAGS -DRSSEG_GLCM_COUNT_UNROLL -o $TMP/k4u.s k4_glcm.hip 2>/dev/null)
hist() { awk -v k="$2" 'index($0,k)==1 && /:/ {p=1} p && $1 ~ /^v_/ {print $1} p && /s_endpgm/ {exit}' "$1" | sort | uniq -c | sort -rn | awk '{print $1, $2}'; }
hist $TMP/k4u.s _Z12k4_glcm_quadPKh > $HERE/${ROUND}_glcm_quad_valu_hist.txt
hist $TMP/k4.s _Z12k4_glcm_pairPKh > $HERE/${ROUND}_glcm_pair_valu_hist.txt
hist $TMP/k4.s _Z14k4_glcm_threadILi7ELi3EE > $HERE/${ROUND}_glcm_thread_7_3_valu_hist.txt
python3 $HERE/valu_mix.py $ROUND
rm -rf $TMP
