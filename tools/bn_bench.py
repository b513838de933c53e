"""BatchNorm row passes, HIP-event timing per launch and TB/s of the bytes they move (bs=512, 96x96):
apply+ReLU (read z, write x'), backward sums (read gy, z), backward apply (read gy, z, write dz) for the 280-channel
layers and for a 70-channel stream slice of the concat buffer.  MMLF_HIP_LIB selects the library build (A/B).
    python tools/bn_bench.py [B] [tag]"""
import os
import sys
import torch
sys.path.insert(0, os.getcwd())
from mmlf_amd import engine, _lib
from mmlf_amd._lib import call, ptr
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
TAG = sys.argv[2] if len(sys.argv) > 2 else os.environ.get('MMLF_HIP_LIB', 'default')
H = W = 96
geo = engine.Geometry(B, H, W)


def t(f, n=10):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def run(C, cs_z, cs_y, c_off, c_store):
    z = torch.randn(geo.alloc * cs_z, device=dev)
    gy = torch.randn(geo.alloc * cs_y, device=dev)
    out = torch.empty(geo.alloc * cs_y, device=dev)
    dz = torch.empty(geo.alloc * cs_z, device=dev)
    coef = torch.rand(4 * C, device=dev) + 0.5
    k = torch.rand(3 * C, device=dev)
    am = torch.zeros(int(_lib.load().mmlf_amax_entries(B, H, W)), device=dev)
    ws = engine._Workspace.get(dev)
    gam = torch.ones(C, device=dev)
    dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    elems = B * H * W * C * 4 / 1e9          # GB of one valid-extent tensor
    a = t(lambda: call('mmlf_bn_apply_relu', ptr(z), cs_z, C, ptr(coef), ptr(coef[C:]), ptr(out), cs_y, c_off, c_store, B, H, W,
                       ptr(am), _lib.stream_ptr()))
    r = t(lambda: call('mmlf_bn_bwd_reduce', ptr(gy), cs_y, c_off, ptr(z), cs_z, C, ptr(coef), ptr(coef[C:]), ptr(gam),
                       ptr(coef[2 * C:]), ptr(coef[3 * C:]), ptr(dg), ptr(db), 1, ptr(k), ptr(ws.partial), engine.BN_BLOCKS,
                       B, H, W, _lib.stream_ptr()))
    b = t(lambda: call('mmlf_bn_bwd_apply', ptr(gy), cs_y, c_off, ptr(z), cs_z, C, ptr(coef), ptr(coef[C:]), ptr(coef[2 * C:]),
                       ptr(k), ptr(dz), cs_z, B, H, W, ptr(am), _lib.stream_ptr()))
    print(f'{TAG} C={C} cs_z={cs_z} cs_y={cs_y}+{c_off}  apply {a:.3f} ms = {2 * elems / a:.2f} TB/s   '
          f'bwd_reduce {r:.3f} ms = {2 * elems / r:.2f} TB/s   bwd_apply {b:.3f} ms = {3 * elems / b:.2f} TB/s', flush=True)


if os.environ.get('BN_BENCH_COPY', '1') != '0':      # what this box's HBM does on a plain device copy of one wide tensor
    src = torch.randn(geo.alloc * 280, device=dev)
    dst = torch.empty_like(src)
    c = t(lambda: dst.copy_(src))
    print(f'{TAG} torch copy of {src.numel() * 4 / 1e9:.2f} GB: {c:.3f} ms = {2 * src.numel() * 4 / 1e9 / c:.2f} TB/s', flush=True)
    del src, dst
run(280, 280, 280, 0, 280)
run(70, 72, 280, 70, 70)
run(70, 72, 72, 0, 72)
