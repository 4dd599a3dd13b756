#!/bin/bash
# Builds liboneprot_hip.so (gfx950) in-tree.  hipcc cross-compiles without a GPU.
# Every translation unit is compiled with -Rpass-analysis=kernel-resource-usage; check_resources.py then fails the build when a product kernel
# instantiation carries scratch (a register spill), so that a spill regression cannot reach a bench run unnoticed.
set -e
cd "$(dirname "$0")"
SRCS="rowops.hip gemm_nt.hip gemm_nt8.hip gemm_nt_ln.hip gemm_tn.hip sgemm.hip attention.hip featops.hip"
OBJS=""
PIDS=""
for s in $SRCS; do
  o="${s%.hip}.o"
  if [ ! -f "$o" ] || [ ! -f "$o.remarks" ] || [ "$s" -nt "$o" ] || [ common.h -nt "$o" ] || [ gemm_epi.h -nt "$o" ] || [ gemm_epi8.h -nt "$o" ] || [ sched_ws.h -nt "$o" ] || [ ../../include/oneprot_hip.h -nt "$o" ]; then
    ( if hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-value -Rpass-analysis=kernel-resource-usage -c "$s" -o "$o.tmp" 2> "$o.remarks.tmp"; then
        mv "$o.tmp" "$o"; mv "$o.remarks.tmp" "$o.remarks"
      else
        grep -v "remark:" "$o.remarks.tmp" >&2; rm -f "$o.remarks.tmp"; exit 1
      fi ) &
    PIDS="$PIDS $!"
    if [ "$s" = gemm_nt8.hip ] || [ "$s" = attention.hip ]; then      # their device ISA as text, for check_async_regs.py (a register written behind the compiler's back)
      ( hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-value -S --cuda-device-only "$s" -o "${s%.hip}.s.tmp" 2> /dev/null && mv "${s%.hip}.s.tmp" "${s%.hip}.s" ) &
      PIDS="$PIDS $!"
    fi
  fi
  OBJS="$OBJS $o"
done
for u in gemm_nt8 attention; do
  if [ ! -f $u.s ]; then
    ( hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-value -S --cuda-device-only $u.hip -o $u.s.tmp 2> /dev/null && mv $u.s.tmp $u.s ) &
    PIDS="$PIDS $!"
  fi
done
for p in $PIDS; do wait $p || { echo "compile failed"; exit 1; }; done
REMARKS=""
for s in $SRCS; do REMARKS="$REMARKS ${s%.hip}.o.remarks"; done
python3 check_resources.py $REMARKS || { echo "build refused: register spills in product kernels (see above)"; exit 1; }
python3 check_async_regs.py gemm_nt8.s attention.s || { echo "build refused: the asynchronously written ticket register of k_gemm8 is copied or reused (see above)"; exit 1; }
hipcc --offload-arch=gfx950 -shared -fPIC -o ../liboneprot_hip.so $OBJS
echo "built $(cd .. && pwd)/liboneprot_hip.so"
# RCCL wrappers (include/oneprot_comm.h) in their own library, so that the kernel library carries no RCCL dependency
if [ ! -f ../liboneprot_comm.so ] || [ comm.cpp -nt ../liboneprot_comm.so ] || [ ../../include/oneprot_comm.h -nt ../liboneprot_comm.so ]; then
  ROCM_LIB="${ROCM_PATH:-/opt/rocm}/lib"
  if [ ! -e "$ROCM_LIB/librccl.so" ]; then
    echo "liboneprot_comm.so NOT built: $ROCM_LIB/librccl.so not found (set ROCM_PATH); the kernel library above is complete, only oneprot_amd.comm.RcclComm is unavailable" >&2
  else
    hipcc -O2 -fPIC -shared -std=c++17 comm.cpp -o ../liboneprot_comm.so -L"$ROCM_LIB" -lrccl
    echo "built $(cd .. && pwd)/liboneprot_comm.so"
  fi
fi
