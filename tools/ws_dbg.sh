# timing experiments on the wave-specialised adjoint (lab library; any HODE_WS_DBG bit below 256 gives WRONG results)
#   1 accumulation waves idle | 2 no W^T products | 4 no mechanistic VJP | 8 no 6-value reduction | 16 s_setprio 2 on the propagation waves | 256 force U = 1 | 512 force U = 2 | 1024 timeline stamps (tools/ws_trace.py)
LAB=$PWD/hybrid-ode-for-glp-1-and-glucose_amd/hode/lab/libhode_lab.so
for d in ${WS_DBG_LIST:-0 256 1 257 2 3 4 8}; do
  echo "== HODE_WS_DBG=$d"
  HODE_LIB=$LAB HODE_WS_DBG=$d timeout -k 10 120 python tools/time_adjoint.py ${WS_B:-4096} 2>&1 | grep adjoint
done
echo "== one-role kernel"
HODE_LIB=$LAB HODE_BWD=fused timeout -k 10 120 python tools/time_adjoint.py ${WS_B:-4096} 2>&1 | grep adjoint
