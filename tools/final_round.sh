#!/bin/bash
# Round-end measurement on the GPU box (run through gpurun): full bench line, rocprofv3 kernel statistics of the
# headline-only bench, two separate PMC passes (FETCH_SIZE, WRITE_SIZE), the committer's phase profile; reduced to
# gpurun_out/final/profiles/<tag>_*.
set -o pipefail
TAG=${1:-v1}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/final
mkdir -p "$O/profiles" && cd /tmp && export TMPDIR=/tmp
echo "[final] committer profile"; python3 "$R/tools/commit_profile.py" "$O/profiles/${TAG}_commit.json" > "$O/commit.log" 2>&1 || echo "(no committer profile)"
cp "$O/profiles/${TAG}_commit.json" "$R/profiles/r03/" 2>/dev/null || true
echo "[final] bench (full)"; python3 -u "$R/bench.py" --steps 20 --warmup 3 > "$O/bench_full.json" 2> "$O/bench_full.err" || exit 1
cp "$O/bench_full.json" "$O/profiles/${TAG}_bench_full.json"
echo "[final] kernel trace"; rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- python3 "$R/bench.py" --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > "$O/bench_prof.json" 2> "$O/bench_prof.err" || exit 2
echo "[final] pmc fetch"; rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/pmc_fetch" -- python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > "$O/pmc_fetch.json" 2> "$O/pmc_fetch.err" || exit 3
echo "[final] pmc write"; rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/pmc_write" -- python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > "$O/pmc_write.json" 2> "$O/pmc_write.err" || exit 4
python3 "$R/tools/profile_summary.py" "$TAG" "$O/stats" "$O/pmc_fetch" "$O/pmc_write" "$O/bench_prof.json" "$O/profiles" || exit 5
echo "[final] done"; tail -c 600 "$O/bench_full.json"
