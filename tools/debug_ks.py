import sys, os, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.filterwarnings("ignore")
import torch, numpy as np
from liberate_fhe_amd.fhe import ckks_engine, presets
from liberate_fhe_amd.utils import synth
name = sys.argv[1] if len(sys.argv) > 1 else "bronze"
eng = ckks_engine(**{**presets.params[name], "devices": ["cuda:0"]})
a = synth.ciphertext(eng, 3, 0)
evk = synth.key_switch_key(eng, 5)
level = 0
d = 0
tabs = eng._ks_tables(level)
N, logN, K = eng.ctx.N, eng.ctx.logN, eng.ntt.num_special_primes
rows, ell = eng._rows(d, level, True), eng._rows(d, level, False)
cs = eng._consts(d, level, True)
src = a.data[1][0]
st = torch.empty_like(src)
nparts_l, desc_d, tab_d = tabs[("digits", d)]
eng.backend.ks_digits(src, st, nparts_l, desc_d, tab_d, eng._consts(d, level, False))
nparts = len(tabs["order"])
desc, E, Ed = tabs[("extend", d)]
key = eng._key_pack(evk)[0]
tw, itw, ninv = eng._tw(d, level, True), eng._tw(d, level, True, True), eng._vec("Ninv", d, level, True)
B = eng.backend
# unfused
ext = torch.empty((nparts, rows, N), dtype=torch.int64, device="cuda")
B.ks_extend(st, ext, nparts, rows, desc, E, cs)
B.ntt(ext, nparts, rows, logN, tw, None, cs)
s_ref = torch.empty((2, rows, N), dtype=torch.int64, device="cuda")
B.ks_inner(ext, key, tabs["first_part"], eng.ntt.starts[level][d], s_ref[0], s_ref[1], nparts, rows, cs)
s_ntt = s_ref.clone()
B.intt(s_ref, 2, rows, logN, itw, ninv, 2, cs)
# fused
for groups in (1, 2):
    tmp = torch.empty((nparts, rows, N), dtype=torch.int64, device="cuda")
    part_s = torch.empty((groups, 2, rows, N), dtype=torch.int64, device="cuda")
    s = torch.zeros((2, rows, N), dtype=torch.int64, device="cuda")
    B.ks_core(st, nparts, rows, logN, desc, E, Ed, key, tabs["first_part"], eng.ntt.starts[level][d], tmp, part_s, groups, s, tw, itw, ninv, cs)
    torch.cuda.synchronize()
    bad = (s != s_ref)
    print(f"groups={groups}: mismatching words {int(bad.sum())} of {bad.numel()}; per (comp,row):", bad.sum(dim=2).tolist())
    # check NTT-domain partial sum residues against the unfused s (before inverse)
    q = torch.tensor([eng.ctx.q[i] for i in eng.ntt.p.destination_arrays_with_special[level][d]], device="cuda")[None, :, None]
    ps = part_s.sum(dim=0) % q
    print("   NTT-domain sums mismatch per (comp,row):", ((ps != (s_ntt % q)).sum(dim=2)).tolist())
    # check tmp (after K2) against pass-1-only? compare ext NTT-domain after fused pass2 is not available; compare tmp residues to unfused pass-1
print("q bits:", [int(x).bit_length() for x in eng.ctx.q])
# --- check the plain extension constants against the Montgomery-form extension on a few samples
R = 1 << 62
dest = eng.ntt.p.destination_arrays_with_special[level][d]
ext_u = torch.empty((nparts, rows, N), dtype=torch.int64, device="cuda")
B.ks_extend(st, ext_u, nparts, rows, desc, E, cs)
stc, extc, Edc, descc = st.cpu().numpy(), ext_u.cpu().numpy(), Ed.cpu().numpy(), desc.cpu().numpy().reshape(-1, 3)
bad = 0
for p in range(nparts):
    r0, alpha, eoff = (int(x) for x in descc[p])
    for r in (0, 3, rows - 1):
        q = eng.ctx.q[dest[r]]
        for j in (0, 5, N - 1):
            plain = sum(int(stc[r0 + i, j]) * int(Edc[eoff + i * rows + r]) for i in range(alpha)) % q
            if (plain * R - int(extc[p, r, j])) % q != 0:
                bad += 1
print("plain-constant check: bad =", bad, " desc =", descc.tolist()[:3], "Ed head", Edc[:12])
# --- isolate: key = 1 everywhere -> fused partial sums must equal sum_p NTT(ext_plain_p) mod q
qv = torch.tensor([eng.ctx.q[i] for i in dest], device="cuda")
ones = torch.ones_like(key)
groups = 1
tmp = torch.empty((nparts, rows, N), dtype=torch.int64, device="cuda")
part_s = torch.empty((groups, 2, rows, N), dtype=torch.int64, device="cuda")
s = torch.zeros((2, rows, N), dtype=torch.int64, device="cuda")
B.ks_core(st, nparts, rows, logN, desc, E, Ed, ones, 0, 0, tmp, part_s, groups, s, tw, itw, ninv, cs)
# reference: plain extension (alpha = 1 for bronze: y_0 mod q_r), exact lf_ntt on it
descl = desc.cpu().numpy().reshape(-1, 3)
assert all(int(a) == 1 for a in descl[:, 1]), "debug path assumes alpha = 1 (bronze)"
extp = torch.stack([torch.remainder(st[int(descl[p, 0])][None, :].expand(rows, N), qv[:, None]) for p in range(nparts)]).contiguous()
B.ntt(extp, nparts, rows, logN, tw, None, cs)
want = torch.remainder(extp, qv[None, :, None]).sum(dim=0) % qv[:, None]
got = part_s[0, 0] % qv[:, None]
print("key=1: NTT-sum mismatches per row:", (got != want).sum(dim=1).tolist())
# and the pass-1 output alone: tmp after K2 should equal pass 1 of lf_ntt on the plain extension?  check residues of
# a full transform done by running only K2 then the stock contiguous pass is not exposed; print a few words instead
print("tmp[0,0,:4]", tmp[0, 0, :4].tolist(), " extp-ntt[0,0,:4] mod q", (extp[0, 0, :4] % qv[0]).tolist(), "part_s[0,0,0,:4]", part_s[0, 0, 0, :4].tolist())
# --- expected pass-1 output for row 0, part 0, coefficients 0..3 (S1 = 2 stages, plain domain)
q0 = eng.ctx.q[dest[0]]
w = [int(v) for v in eng.ctx.psi_br[dest[0]][:4]]
xin = [int(v) for v in torch.remainder(st[int(descl[0, 0])], q0).cpu().tolist()]
n = N
def bf(u, v, tw): return ((u + tw * v) % q0, (u - tw * v) % q0)
a_ = [xin[j] for j in (0, 1, 2, 3)] ; b_ = [xin[j + n // 2] for j in (0, 1, 2, 3)]
c_ = [xin[j + n // 4] for j in (0, 1, 2, 3)] ; d_ = [xin[j + n // 4 + n // 2] for j in (0, 1, 2, 3)]
top = [bf(a_[i], b_[i], w[1])[0] for i in range(4)]
mid = [bf(c_[i], d_[i], w[1])[0] for i in range(4)]
exp = [bf(top[i], mid[i], w[2])[0] for i in range(4)]
print("expected pass-1 words:", exp, " fused tmp:", tmp[0, 0, :4].tolist())
# --- single digit, unit key: part_s must be NTT(ext_plain_0)
for np_ in (1, 2):
    part_s1 = torch.zeros((1, 2, rows, N), dtype=torch.int64, device="cuda")
    B.ks_core(st, np_, rows, logN, desc, E, Ed, ones, 0, 0, tmp, part_s1, 1, s, tw, itw, ninv, cs)
    want1 = torch.remainder(extp[:np_], qv[None, :, None]).sum(dim=0) % qv[:, None]
    got1 = part_s1[0, 0] % qv[:, None]
    print(f"nparts={np_} unit key mismatches per row:", (got1 != want1).sum(dim=1).tolist())
    if np_ == 1:
        bad = (got1[0] != want1[0]).nonzero().flatten()
        print("  first bad idx:", bad[:8].tolist(), " count", len(bad), " got", got1[0, :4].tolist(), "want", want1[0, :4].tolist())
