#!/bin/bash
# Diagnostic only (not product code): builds decoder.hip variants with parts ablated and times the
# backward kernel alone on a cfg-4-like edge list.  Usage on the GPU box: bash tools/ablate_decoder.sh
set -e
cd "$(dirname "$0")/.."
OUT=gpurun_out/ablate; mkdir -p $OUT
for v in base nomfma; do
  flags=""
  
  
  [ $v = nomfma ] && flags="-DPANGNN_ABLATE_MFMA"
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $flags -shared pangnn_amd/csrc/decoder.hip pangnn_amd/csrc/edge_ops.hip -o $OUT/libdec_$v.so
done
python - <<'PY'
import ctypes as C, torch, time
torch.manual_seed(0)
dev=torch.device('cuda')
n=1_000_000; e=74_000_000
gen=torch.Generator(device=dev).manual_seed(0)
src=torch.arange(e,device=dev)//74            # sorted by source like the bench graph
dst=(src//50000*50000 + 50000 + torch.randint(0,50000,(e,),device=dev,generator=gen)).clamp_(max=n-1)
ei=torch.stack([src,dst]).contiguous()
pq=torch.randn(n,128,device=dev)
w2=torch.randn(64,64,device=dev)/8; b2=torch.randn(64,device=dev); w3=torch.randn(64,device=dev); b3=torch.randn(1,device=dev)
g=torch.randn(e,device=dev)*1e-8
gh1=torch.empty(e,64,device=dev); gw2=torch.empty(64,64,device=dev); gb2=torch.empty(64,device=dev); gw3=torch.empty(64,device=dev); gb3=torch.empty(1,device=dev)
logits=torch.empty(e,device=dev)
P=C.c_void_p
for v in ['base','nomfma']:
    lib=C.CDLL(f'gpurun_out/ablate/libdec_{v}.so')
    lib.pangnn_decoder_mlp_bwd_workspace_bytes.restype=C.c_size_t
    wsb=lib.pangnn_decoder_mlp_bwd_workspace_bytes(C.c_int64(e))
    ws=torch.empty(wsb,dtype=torch.uint8,device=dev)
    def bwd():
        return lib.pangnn_decoder_mlp_bwd_f32(P(pq.data_ptr()),C.c_int64(128),P(pq.data_ptr()+256),C.c_int64(128),C.c_int64(n),P(ei.data_ptr()),C.c_int64(e),C.c_int64(e),None,None,P(w2.data_ptr()),P(b2.data_ptr()),P(w3.data_ptr()),P(b3.data_ptr()),C.c_int32(64),P(g.data_ptr()),P(gh1.data_ptr()),P(gw2.data_ptr()),P(gb2.data_ptr()),P(gw3.data_ptr()),P(gb3.data_ptr()),None,P(ws.data_ptr()),C.c_size_t(wsb),None)
    def fwd():
        return lib.pangnn_decoder_mlp_fwd_f32(P(pq.data_ptr()),C.c_int64(128),P(pq.data_ptr()+256),C.c_int64(128),C.c_int64(n),P(ei.data_ptr()),C.c_int64(e),C.c_int64(e),None,None,P(w2.data_ptr()),P(b2.data_ptr()),P(w3.data_ptr()),P(b3.data_ptr()),C.c_int32(64),P(logits.data_ptr()),None)
    for name,fn in (('bwd',bwd),('fwd',fwd)):
        assert fn()==0
        torch.cuda.synchronize(); t0=time.perf_counter()
        for _ in range(5): fn()
        torch.cuda.synchronize(); print(v,name,(time.perf_counter()-t0)/5*1e3,'ms')
PY
