#!/usr/bin/env bash
# tools/build_variant.sh <variant.hip> <name> -- link build/variants/libmhx_<name>.so from a variant of
# minhash_kernels.hip plus the current objects of the other sources (A/B runs: MHX_LIBRARY=... bench.py).
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
SRC="$1"; NAME="$2"
OBJ="${ROOT}/build/mhx"; OUT="${ROOT}/build/variants"; mkdir -p "${OUT}"
bash "${ROOT}/datasketch_amd/csrc/build.sh" > /dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off \
  -I"${ROOT}/include" -I"${ROOT}/datasketch_amd/csrc" -Wall -Wno-unused-function -c "${SRC}" -o "${OUT}/minhash_${NAME}.o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "${OUT}/libmhx_${NAME}.so" "${OBJ}/mhx_api.o" "${OUT}/minhash_${NAME}.o" \
  "${OBJ}/weighted_kernels.o" "${OBJ}/pack_kernels.o" "${OBJ}/sha1_kernels.o" "${OBJ}/comm.o" -ldl
echo "${OUT}/libmhx_${NAME}.so"
