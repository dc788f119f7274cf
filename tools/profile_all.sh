#!/bin/bash
# tools/profile_all.sh [PREFIX] -- on the GPU box: one rocprofv3 summary (kernel trace + stats, FETCH_SIZE, WRITE_SIZE in separate
# passes: tools/profile_gpu.sh) per BASELINE configuration and north_star workload, all with the library this snapshot carries,
# so that every bench line's `traffic` -- the headline's and the companions' -- refers to the device code that is benched
# (VERDICT r04 item 4).  Summaries land in gpurun_out/PREFIX_<name>_summary.{md,json}; copy them to profiles/.
PREFIX=${1:-r06_prof}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
LEAN="--no-companions --no-config3 --no-north-star --no-host-boundary --no-cold --no-drop-in"
run() { name=$1; shift; echo "== $name: $*"; bash tools/profile_gpu.sh ${PREFIX}_${name} $LEAN "$@" > gpurun_out/${PREFIX}_${name}.log 2>&1 || { echo "$name FAILED"; tail -5 gpurun_out/${PREFIX}_${name}.log; }; grep -E "^\| .*(csr_|coo_|ell_)" gpurun_out/${PREFIX}_${name}_summary.md | head -4; }
run poisson_csr_final
run queen_csr_final --workload queen
run kkt_csr_final --workload kkt
# round 6: the stored lower triangles (the reference's default semantics for the symmetric SuiteSparse files)
run queen_stored_csr_final --workload queen_stored
run kkt_stored_csr_final --workload kkt_stored
run webbase_csr_final --workload webbase
run webbase_coo_final --workload webbase --format coo
run webbase_hybrid_final --workload webbase --format hybrid
run banded_csr_final --workload banded
run random24_csr_final --workload random24
