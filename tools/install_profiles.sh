#!/bin/bash
# Copy what tools/profile_round.sh and tools/refresh_profiles.sh left under gpurun_out/<tag>/ into
# profiles/ under the round's prefix (run in the build container after the gpurun call):
#   bash tools/install_profiles.sh r02f r02
TAG=${1:?tag under gpurun_out/}; PFX=${2:-r03}    # the tag directory must be fresh: everything in it is copied
cd "$(dirname "$0")/.."
S=gpurun_out/$TAG
for w in C F D; do
  [ -f $S/traffic_config$w.json ] && cp $S/traffic_config$w.json profiles/traffic_config$w.json
  for f in config${w}_bench_kernel_stats.csv config${w}_fetch_summary.json config${w}_write_summary.json config${w}_bench_ktrace_summary.json; do
    [ -f $S/$f ] && cp $S/$f profiles/${PFX}_$f
  done
done
# a bench line goes to profiles/ only if it parses and its own output check passed
for f in $S/bench_*.json; do
  [ -s "$f" ] || continue
  if python3 - "$f" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
v = j.get("verify")
sys.exit(0 if (v is None or v.get("ok") is True) and j.get("value") else 1)
PY
  then cp "$f" profiles/${PFX}_$(basename $f)
  else echo "REFUSED $f: no value, or verify.ok is not true"; BAD=1
  fi
done
python3 - <<'PY'
import glob, json, os, sys
sys.path.insert(0, os.getcwd())
import bench
h = bench.source_hash()
for f in sorted(glob.glob("profiles/traffic_config*.json")):
    print(f, json.load(open(f))["source_hash"], "(sources now %s)" % h)
for f in sorted(glob.glob("profiles/r03_bench_config[CFD].json")):
    j = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, "ms %.4f" % j["ms_per_step"], "frac %.3f" % j["roofline"]["frac"], "traffic_stale", j["roofline"]["traffic_stale"],
          "verify", j.get("verify", {}).get("ok"))
PY
[ -z "$BAD" ] || { echo "install_profiles: at least one line was refused"; exit 1; }
