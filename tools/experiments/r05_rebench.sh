# bench line + kernel trace again after a kernel change that does not need the whole refresh (outputs where tools/keep_r05.sh <tag> looks for them)
TAG="${1:-r05d}"; SRC="${2:-r05c}"
OUT="gpurun_out/refresh_${TAG}"; mkdir -p "${OUT}"
timeout 300 python -m pytest tests -q -m gpu -p no:cacheprovider -k "bucket or sort or lsh or candidate or query or bench" > "${OUT}/pytest_part.log" 2>&1; echo "pytest(part) rc=$?"; tail -1 "${OUT}/pytest_part.log"
timeout 900 python bench.py > "${OUT}/bench.log" 2>&1; echo "bench rc=$?"; tail -1 "${OUT}/bench.log" > "${OUT}/bench.json"; cut -c1-200 "${OUT}/bench.json"
PROFILE_ONLY=trace timeout 600 bash tools/profile.sh "${TAG}" > "${OUT}/profile.log" 2>&1; echo "profile rc=$?"
python tools/rocpd_summary.py "gpurun_out/prof_${TAG}" > "${OUT}/rocprof_summary.txt" 2>&1 || true
find gpurun_out -name "*.db" -delete 2>/dev/null; find gpurun_out -type f -size +8M -delete 2>/dev/null; du -sh gpurun_out | tail -1
