#!/bin/bash
# Is there any way to produce tantivy-written bytes on the GPU box?
# Output: gpurun_out/toolchain_probe.log  (committed as profiles/r02_toolchain_probe.log)
R=${GRAFT_REPO_ROOT:-/root/repo}
L=$R/gpurun_out/toolchain_probe.log
mkdir -p $R/gpurun_out
{
  echo "== date"; date -u
  echo "== cargo / rustc"; (cargo --version; rustc --version) 2>&1
  echo "== which"; which cargo rustc rustup 2>&1
  echo "== ~/.cargo, ~/.rustup"; ls -d ~/.cargo ~/.rustup /usr/local/cargo /opt/rust* 2>&1
  echo "== python -c 'import tantivy'"; python -c 'import tantivy; print(tantivy.__version__)' 2>&1 | tail -1
  echo "== pip download tantivy (5 s timeout, no network expected)"; timeout 20 pip download --no-deps -d /tmp/tv tantivy 2>&1 | tail -3
  echo "== pip index / wheelhouse"; pip config list 2>&1; ls /opt/wheelhouse /wheelhouse /root/wheelhouse 2>&1 | head
  echo "== any *.crate / bitpacking sources on disk"; find / -xdev \( -name "*.crate" -o -iname "bitpacking*" \) -not -path "/proc/*" 2>/dev/null | head
  echo "== site-packages shared objects mentioning bitpacking"; grep -l "bitpacking" $(python -c 'import site; print(" ".join(site.getsitepackages()))')/*/*.so 2>/dev/null | head
  echo "== nproc / affinity / cgroup"; nproc; python -c 'import os; print(len(os.sched_getaffinity(0)), os.cpu_count())'; cat /sys/fs/cgroup/cpu.max 2>&1; cat /proc/loadavg
  echo "== rocm-smi"; rocm-smi --showmeminfo vram 2>&1 | tail -5
} > $L 2>&1
cat $L
