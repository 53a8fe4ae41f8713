"""Check the EPI_L2MIN planes against a torch reference of the same bf16 distance matrix."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vtc_amd import _lib as L
lib = L.lib()
f = lib.vtc_debug_l2min
f.restype = C.c_int
f.argtypes = [C.c_void_p] * 4 + [C.c_int] * 4 + [C.c_void_p] * 3
nb, na, d = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (1000, 777, 128)))
for rb in (128, 64):
    g = torch.Generator().manual_seed(1)
    b = torch.nn.functional.normalize(torch.randn(nb, d, generator=g), dim=-1).cuda()
    a = torch.nn.functional.normalize(torch.randn(na, d, generator=g), dim=-1).cuda()
    qb, gb = b.bfloat16().contiguous(), a.bfloat16().contiguous()
    qn, gn = (b * b).sum(1).contiguous(), (a * a).sum(1).contiguous()
    nbc, nbr = (na + 63) // 64, (nb + rb - 1) // rb
    rowk = torch.full((3, nbc, nb), 7, dtype=torch.int32, device="cuda")
    colk = torch.full((3, nbr, na), 7, dtype=torch.int32, device="cuda")
    rc = f(qb.data_ptr(), gb.data_ptr(), qn.data_ptr(), gn.data_ptr(), nb, na, d, rb, rowk.data_ptr(), colk.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0, lib.vtc_last_error()
    torch.cuda.synchronize()
    D = (qn[:, None] - 2.0 * (qb.float() @ gb.float().t())) + gn[None, :]
    D = D.clamp_min(0)
    bits = (D.view(torch.int32) & ~127).long()
    # rows: per (row, 64-col block): keys = bits | col-in-block, sorted ascending, first 3
    pad = nbc * 64 - na
    Kr = torch.nn.functional.pad(bits | (torch.arange(na, device="cuda") % 64)[None, :], (0, pad), value=0xFFFFFFFF).reshape(nb, nbc, 64)
    ref = Kr.sort(dim=2).values[:, :, :3].permute(2, 1, 0)                    # [3, nbc, nb]
    got = rowk.long() & 0xFFFFFFFF
    # fp32 accumulation order differs from torch's: compare distances loosely, indices exactly where unambiguous
    dref, dgot = (ref & ~127).int().view(torch.float32), (got & ~127).int().view(torch.float32)
    ok = (ref != 0xFFFFFFFF)
    print(f"rb={rb} rows: max |d| diff {float((dref - dgot)[ok].abs().max()):.2e}; idx equal frac {float(((ref & 127) == (got & 127))[ok].float().mean()):.5f}; "
          f"inf pattern equal {bool(((ref == 0xFFFFFFFF) == (got == 0xFFFFFFFF)).all())}")
    padr = nbr * rb - nb
    Kc = torch.nn.functional.pad((bits | (torch.arange(nb, device="cuda") % rb)[:, None]).t(), (0, padr), value=0xFFFFFFFF).reshape(na, nbr, rb)
    refc = Kc.sort(dim=2).values[:, :, :3].permute(2, 1, 0)                   # [3, nbr, na]
    gotc = colk.long() & 0xFFFFFFFF
    drefc, dgotc = (refc & ~127).int().view(torch.float32), (gotc & ~127).int().view(torch.float32)
    okc = (refc != 0xFFFFFFFF)
    print(f"rb={rb} cols: max |d| diff {float((drefc - dgotc)[okc].abs().max()):.2e}; idx equal frac {float(((refc & 127) == (gotc & 127))[okc].float().mean()):.5f}; "
          f"inf pattern equal {bool(((refc == 0xFFFFFFFF) == (gotc == 0xFFFFFFFF)).all())}")
    for pl in range(3):
        bad = ((refc[pl] & 127) != (gotc[pl] & 127)) & okc[pl]
        print("   col plane", pl, "idx mismatches", int(bad.sum()), "of", int(okc[pl].sum()), "| row plane", pl, int((((ref[pl] & 127) != (got[pl] & 127)) & ok[pl]).sum()))
    import collections
    bad = (((ref[1] & 127) != (got[1] & 127)) & ok[1]).nonzero()
    if bad.numel():
        blks = collections.Counter(bad[:, 0].tolist()); rows_ = bad[:, 1]
        print("   row-plane-1 mismatches by col block:", dict(list(blks.items())[:12]), "rows min/max", int(rows_.min()), int(rows_.max()),
              "rows%256 sample", sorted(set((rows_ % 256).tolist()))[:20], "example got/ref", hex(int(got[1][bad[0,0], bad[0,1]])), hex(int(ref[1][bad[0,0], bad[0,1]])), hex(int(got[0][bad[0,0], bad[0,1]])))
