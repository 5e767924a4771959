# The first 8-GPU run in ONE call: every GPU count x exchange form, efficiency + overlap per cell, the winner.
#   bash tools/scale_check.sh             on an N-GPU node (cells beyond the box's GPU count are skipped)
#   bash tools/scale_check.sh --dry-run   two ranks sharing one GPU, tiny shapes: the code path of every cell
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cd "$(dirname "$0")/.." && python tools/scale_check.py "$@"
