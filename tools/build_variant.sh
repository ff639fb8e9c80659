#!/usr/bin/env bash
# tools/build_variant.sh <variant.hip> <name> [module] -- link build/variants/libmhx_<name>.so from a variant
# of one kernel source (module = minhash_kernels by default, or weighted_kernels / pack_kernels / sha1_kernels)
# plus the current objects of the other sources (A/B runs: MHX_LIBRARY=... bench.py, tools/ab.sh).
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
SRC="$1"; NAME="$2"; MOD="${3:-minhash_kernels}"
OBJ="${ROOT}/build/mhx"; OUT="${ROOT}/build/variants"; mkdir -p "${OUT}"
bash "${ROOT}/datasketch_amd/csrc/build.sh" > /dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off \
  -I"${ROOT}/include" -I"${ROOT}/datasketch_amd/csrc" -Wall -Wno-unused-function -c "${SRC}" -o "${OUT}/${MOD}_${NAME}.o"
OBJS=()
for m in mhx_api minhash_kernels weighted_kernels pack_kernels sha1_kernels lsh_kernels comm; do
  if [[ "$m" == "$MOD" ]]; then OBJS+=("${OUT}/${MOD}_${NAME}.o"); else OBJS+=("${OBJ}/${m}.o"); fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "${OUT}/libmhx_${NAME}.so" "${OBJS[@]}" -ldl
echo "${OUT}/libmhx_${NAME}.so"
