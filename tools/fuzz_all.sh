#!/bin/bash
# Every mode of tools/fuzz_vs_reference.py once, with ONE fresh seed that is recorded in the log (the
# committed logs of r03-r05 were all seeds 0-4; a seed nobody had run found a bug).  Default seed = the commit count.
#   tools/fuzz_all.sh [seed] [cases] [tag]      ->  profiles/<tag>_fuzz_<mode>_seed<seed>.txt     (build container only)
SEED=${1:-$(git rev-list --count HEAD)}
CASES=${2:-100}
TAG=${3:-r06}
cd "$(dirname "$0")/.."
export PYTHONDONTWRITEBYTECODE=1
run() {
  m=$1; shift
  out=profiles/${TAG}_fuzz_${m}_seed${SEED}.txt
  { echo "# tools/fuzz_vs_reference.py $m $SEED $CASES  ($(git rev-parse --short HEAD), $*)"; env "$@" timeout 3000 python tools/fuzz_vs_reference.py $m $SEED $CASES 2>&1 | tail -40; } > $out
  echo "$m: $(tail -1 $out)"
}
for m in fixed adaptive adjoint backprop; do run $m TDEQ_FUZZ_X=1 & done; wait
for m in event tableau eventgrad; do run $m TDEQ_FUZZ_X=1 & done
run complex TDEQ_FUZZ_BACKEND=host &      # (the C oracle has no complex kernels: this mode runs on the package's torch-op host path)
wait
for m in callbacks hessian vectol brow; do run $m TDEQ_FUZZ_X=1 & done; wait
run hostexact TDEQ_FUZZ_X=1
