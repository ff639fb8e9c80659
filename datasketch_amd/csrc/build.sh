#!/usr/bin/env bash
# Build libmhx.so for gfx950 (MI355X) in-tree: datasketch_amd/libmhx.so
# hipcc cross-compiles without a GPU.  -ffp-contract=off: the weighted path must not fuse a*b+c.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="${HERE}/../libmhx.so"
OBJ="${MHX_OBJ_DIR:-${HERE}/../../build/mhx}"
mkdir -p "${OBJ}"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off
       -I"${HERE}/../../include" -I"${HERE}" -Wall -Wno-unused-function)
pids=()
for src in mhx_api minhash_kernels weighted_kernels pack_kernels sha1_kernels lsh_kernels comm; do
  if [[ ! -f "${OBJ}/${src}.o" || "${HERE}/${src}.hip" -nt "${OBJ}/${src}.o" \
        || "${HERE}/mhx_internal.h" -nt "${OBJ}/${src}.o" || "${HERE}/band_digest.h" -nt "${OBJ}/${src}.o" || "${HERE}/../../include/mhx.h" -nt "${OBJ}/${src}.o" ]]; then
    "${HIPCC}" "${FLAGS[@]}" -c "${HERE}/${src}.hip" -o "${OBJ}/${src}.o" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [[ -n "${p}" ]] && wait "${p}"; done
"${HIPCC}" --offload-arch=gfx950 -shared -fPIC -o "${OUT}" "${OBJ}"/mhx_api.o "${OBJ}"/minhash_kernels.o \
  "${OBJ}"/weighted_kernels.o "${OBJ}"/pack_kernels.o "${OBJ}"/sha1_kernels.o "${OBJ}"/lsh_kernels.o "${OBJ}"/comm.o -ldl
echo "built ${OUT}"
# CPython helper that packs Python byte tokens (host glue, plain C)
PYINC="$(python3 -c 'import sysconfig; print(sysconfig.get_paths()["include"])')"
gcc -O2 -shared -fPIC -Wall -I"${PYINC}" "${HERE}/pack_module.c" -o "${HERE}/../_mhxpack.so"
echo "built ${HERE}/../_mhxpack.so"
