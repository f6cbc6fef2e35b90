#!/bin/bash
# run a command; if a GPU core dump appears, print where the faulting waves were
cd ${GRAFT_REPO_ROOT:-$(pwd)}
rm -f gpucore.*
"$@" > gpurun_out/core_run.log 2>&1
echo "exit $?" >> gpurun_out/core_run.log
C=$(ls gpucore.* 2>/dev/null | head -1)
if [ -n "$C" ]; then
  ls -la $C >> gpurun_out/core_run.log
  timeout 300 /opt/rocm/bin/rocgdb -batch -ex "info threads" /usr/bin/python3.10 --core=$C > gpurun_out/core_threads.txt 2>&1
  grep -c "AMDGPU Wave" gpurun_out/core_threads.txt
  grep "AMDGPU Wave" gpurun_out/core_threads.txt | sed 's/.*) //' | cut -c1-160 | sort | uniq -c | sort -rn | head -20
  grep -i "exception\|fault\|signal\|viol" gpurun_out/core_threads.txt | head
else
  echo "no core"; tail -3 gpurun_out/core_run.log
fi
