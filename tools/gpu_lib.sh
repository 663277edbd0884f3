# Sourced by the tools/gpu_*.sh scripts.  tos SECONDS cmd...: run one GPU step under `timeout -k 10`; if it hits the limit
# (a hung kernel, a dead box) the whole call stops there with exit code 9 - no further GPU step is started after a timeout.
tos() {
  local lim=$1; shift
  timeout -k 10 "$lim" "$@"
  local rc=$?
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then
    echo "[stop] a GPU step hit its ${lim}s limit ($1 ...): no further GPU step in this call" >&2
    exit 9
  fi
  return $rc
}
# sub SCRIPT args...: run another gpu_*.sh and stop the call if it stopped for that reason
sub() { "$@"; local rc=$?; [ $rc -eq 9 ] && exit 9; return $rc; }
