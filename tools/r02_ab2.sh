#!/bin/bash
set -u
V=liberate_fhe_amd/csrc/variants
python -m pytest tests/test_keygen_golden.py tests/test_ntt_cuda_gpu.py -m gpu -x -q 2>&1 | tail -2
for i in 1 2; do
  for L in liberate_fhe_amd/csrc/libckks_hip.so $V/lib_pipe2.so $V/lib_pipe4.so $V/lib_pipe8.so; do
    LF_HIP_LIB=$PWD/$L python bench.py --no-extra 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$L', round(d['value']), 'step_ms', round(d['ms_per_step'],3), 'tiled_us', round(r['avg_launch_ms']*1e3,1), 'cols_us', round(r['column_pass_launch_ms']*1e3,1))"
  done
done
# correctness of the pipelined variant on a big batch: compare with the default library bit for bit
LF_HIP_LIB=$PWD/$V/lib_pipe4.so python - <<'PY'
import numpy as np, torch, sys
sys.path.insert(0, '.')
from liberate_fhe_amd._native import lib, check
from liberate_fhe_amd.ntt import twiddles, ntt_context
from liberate_fhe_amd.fhe.context.ckks_context import ckks_context
from liberate_fhe_amd.utils import synth
import ctypes
ctx = ckks_context(logN=16, num_special_primes=4); ntt = ntt_context(ctx, devices=["cuda:0"])
L=30; tot=len(ctx.q); lo=tot-L; rows=list(range(lo,tot)); B=64
x = torch.stack([torch.from_numpy(synth.uniform_rows(b, rows, ctx.q, ctx.N, lazy=True)) for b in range(B)]).cuda()
sl=lambda t:t[0][lo:]
psi,q2,ql,qh,kl,kh=(sl(t) for t in (ntt.psi,ntt._2q,ntt.ql,ntt.qh,ntt.kl,ntt.kh))
st=torch.cuda.current_stream().cuda_stream
dp=twiddles.dp_pointer(psi,ql,qh,kl,kh,0,st); qh_=np.array([ctx.q[i] for i in rows],dtype=np.int64)
ref = ctypes.CDLL("liberate_fhe_amd/csrc/libckks_hip.so")
def run(l, t): 
    f=l.lf_ntt; f.argtypes=lib.lf_ntt.argtypes; f.restype=ctypes.c_int
    assert f(t.data_ptr(),B,L,16,psi.data_ptr(),dp,qh_.ctypes.data,0,0,q2.data_ptr(),ql.data_ptr(),qh.data_ptr(),kl.data_ptr(),kh.data_ptr(),0,st)==0
a=x.clone(); b=x.clone(); run(lib,a); run(ref,b); torch.cuda.synchronize()
print("pipelined == plain:", torch.equal(a,b))
PY
