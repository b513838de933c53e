"""Host-side orchestration of the HIP kernels for the FeedForward trunk
(reference mmlf/model/feed_forward.py:206-269 forward; autograd backward reached from
mmlf/train/cli.py:257).  PyTorch supplies device memory and streams only; every arithmetic
step is a call through the C ABI (include/mmlf_hip.h).

Layout and indexing are described in include/mmlf_hip.h and DESIGN.md section 3.
"""
import os
import threading

import torch

from . import _lib
from ._lib import call, ptr

VAR_IDENTITY, VAR_TRANSPOSE, VAR_TRANSPOSE_FLIPH = 0, 1, 2
BN_BLOCKS = 1024
LOSS_BLOCKS = 1024


def cs_of(c):
    """channel stride (floats) used for a c-channel grid tensor: multiple of 8."""
    return (c + 7) // 8 * 8


class Geometry:
    def __init__(self, B, H, W):
        self.B, self.H, self.W = B, H, W
        # pitch W + 2, H + 2 rows (the library's build options MMLF_GRID_PAD_W / _H: csrc/common.h has the measured alternatives)
        self.P, self.R = W + int(_lib.load().mmlf_grid_pad_w()), H + int(_lib.load().mmlf_grid_pad_h())
        self.G = self.P * self.R
        self.NQ = B * self.G
        self.alloc = int(_lib.load().mmlf_grid_alloc_positions(B, H, W))
        self.amax_n = int(_lib.load().mmlf_amax_entries(B, H, W))
        self.amax_head = int(_lib.load().mmlf_amax_head())             # tensor-maximum shards, then one entry per grid row
        self.amax_stride = int(_lib.load().mmlf_amax_shard_stride())
        if self.alloc * 288 * 4 >= 2 ** 63 or self.NQ + 2 * self.P + 600 >= 2 ** 31:
            raise ValueError('batch x image too large for 32-bit grid positions')

    def buf(self, cs, device):
        """Grid buffer with zeroed head/tail slack (the kernels write everything else) and its zeroed amax
        array: a head of 64 partial maxima of |x| over the tensor (`amax_stride` floats apart), then
        [amax_head + r] = max |x| of grid row r; the tensor's producers raise the entries by atomic max and the
        f16-split kernels derive their power-of-two operand scales from them (include/mmlf_hip.h)."""
        t = torch.empty(self.alloc * cs, dtype=torch.float32, device=device)
        t.absmax = torch.empty(self.amax_n, dtype=torch.float32, device=device)
        call('mmlf_zero_slack', ptr(t), cs, self.B, self.H, self.W, ptr(t.absmax), _lib.stream_ptr())
        return t

    def bufs(self, css, device):
        """several grid buffers (channel strides `css`, at most four) whose slack and amax arrays ONE launch zeroes"""
        assert 1 <= len(css) <= 4
        if not BATCHED:
            return [self.buf(cs, device) for cs in css]
        ts = []
        for cs in css:
            t = torch.empty(self.alloc * cs, dtype=torch.float32, device=device)
            t.absmax = torch.empty(self.amax_n, dtype=torch.float32, device=device)
            ts.append(t)
        import ctypes
        n = len(ts)
        grid = (ctypes.c_void_p * 4)(*([ptr(t) for t in ts] + [None] * (4 - n)))
        amax = (ctypes.c_void_p * 4)(*([ptr(t.absmax) for t in ts] + [None] * (4 - n)))
        csa = (ctypes.c_int * 4)(*(list(css) + [0] * (4 - n)))
        call('mmlf_zero_slack4', grid, csa, amax, self.B, self.H, self.W, _lib.stream_ptr())
        return ts

    def relu_mask(self, device):
        """words for the bit form of one layer's ReLU mask (mmlf_conv2x2_h2 relu_mask_out / relu_mask_in)"""
        n = int(_lib.load().mmlf_relu_mask_words(self.B, self.H, self.W))
        return torch.empty(n, dtype=torch.int32, device=device)

    def amax_of(self, t, cs):
        """amax array of a grid tensor that did not come from buf() (tests, tools): computed with torch ops."""
        rows = t[:self.NQ * cs].view(self.B * self.R, self.P * cs).abs().amax(1)
        out = torch.zeros(self.amax_n, dtype=torch.float32, device=t.device)
        out[0] = rows.max()                # one shard holds it all
        out[self.amax_head:self.amax_head + rows.numel()] = rows
        return out

    def amax_canonical(self, a):
        """an amax array with the tensor maximum collapsed into shard 0 (what amax_of builds): kernels spread it over
        the shards by wave / row, so two arrays that describe the same tensor compare equal in this form only"""
        out = a.clone()
        out[:self.amax_head] = 0
        out[0] = a[:self.amax_head].max()
        return out


class _Workspace:
    """Scratch that is reused across calls (stream-ordered: one stream at a time, `enter_stream` orders a change of
    stream), one per (device, calling thread):
    nn.DataParallel drives replicas from one thread per device -- and nothing stops two of them from sharing a
    device -- while autograd's backward runs on its own thread per device.  Kept in thread-local storage, so a
    workspace (an 8 MB partial-sum buffer, a side stream, up to a few hundred MB of weight-gradient scratch) dies
    with its thread: DataParallel.parallel_apply starts fresh threads on every forward."""
    _tls = threading.local()

    @classmethod
    def get(cls, device):
        table = cls._tls.__dict__.setdefault('table', {})
        key = (device.type, device.index)
        ws = table.get(key)
        if ws is None:
            ws = table[key] = cls(device)
        return ws

    def __init__(self, device):
        self.device = device
        self.wgrad = None
        self.wgrad_side = None                       # workspace of weight-gradient launches on the side stream
        self.side = torch.cuda.Stream(device=device) if device.type == 'cuda' else None
        self.partial = torch.empty(2 * 512 * max(BN_BLOCKS, LOSS_BLOCKS) + 8, dtype=torch.float64, device=device)

    def packed_filters(self, items):
        """f16-split packed forms of many filters from ONE launch (mmlf_pack_filters_h2).  items: list of
        (key, weight tensor (Cout, Cin, 2, 2), variant, dgrad).  The descriptor table and the packed buffers persist
        across steps as long as the same weight storage is passed (Adam updates weights in place); the contents are
        re-made on every call.  Returns {key: packed tensor}.

        Conditions of use (the views are handed to the tape and read again by backward):
          * the weights must not change between a forward pass and its backward pass (the packed data-gradient forms
            were made from the forward's weights; the reference's loop and TrainStep step the optimizer after backward);
          * one (thread, device) workspace serves ONE stream at a time: the store is overwritten in stream order by the
            next forward.  If the calling thread switches streams, the new stream first waits for an event recorded on the
            old one behind this workspace's last use (`_last_use`), so a repack cannot overtake convolutions still reading
            the store;
          * one cache entry per signature (training packs forward + data-gradient forms, evaluation forward forms only):
            alternating train and eval steps re-uses both instead of re-allocating and re-uploading the table."""
        import numpy as np
        lib = _lib.load()
        sig = tuple((w.data_ptr(), w.shape[0], w.shape[1], int(var), int(dg)) for _, w, var, dg in items)
        self.enter_stream()
        caches = self.__dict__.setdefault('_packs', {})
        cache = caches.get(sig)
        if cache is None:
            if len(caches) >= 4:                       # (weights re-allocated: drop the stale entries)
                caches.clear()
            desc = np.zeros(len(items), dtype=np.dtype([('w', '<u8'), ('packed', '<u8'), ('Cout', '<i4'), ('Cin', '<i4'),
                                                         ('variant', '<i4'), ('dgrad', '<i4'), ('col0', '<i4'), ('np', '<i4')]))
            offs, total, col = [], 0, 0
            for i, (_, w, var, dg) in enumerate(items):
                cout, cin = w.shape[0], w.shape[1]
                K, N = (cout, cin) if dg else (cin, cout)
                nbytes = int(lib.mmlf_packed_filter_h2_bytes(cs_of(K), N))
                npk = int(lib.mmlf_packed_filter_h2_columns(N))
                if nbytes < 0 or npk < 0:
                    raise RuntimeError(f'pack_filters: unsupported channels K={K} N={N}')
                offs.append((total, nbytes))
                desc[i] = (w.data_ptr(), 0, cout, cin, int(var), int(dg), col, npk)
                total += (nbytes + 255) // 256 * 256
                col += npk
            store = torch.empty(total, dtype=torch.uint8, device=self.device)
            for i, (o, _) in enumerate(offs):
                desc['packed'][i] = store.data_ptr() + o
            table = torch.from_numpy(desc.view(np.uint8).copy()).to(self.device)
            views = [store[o:o + n].view(torch.float32) for o, n in offs]
            cache = caches[sig] = (sig, table, store, views, col)
        _, table, _, views, col = cache
        call('mmlf_pack_filters_h2', ptr(table), len(items), col, _lib.stream_ptr())
        return {key: v for (key, _, _, _), v in zip(items, views)}

    def enter_stream(self):
        """Orders a change of stream for EVERY buffer this workspace owns (the packed-filter store, `partial`, the
        scratch and weight-gradient buffers): if the calling thread has moved to another stream since the workspace was
        last used, the new stream first waits for the work the previous one had enqueued.  Called from the places that hand
        out or overwrite those buffers -- packed_filters(), scratch(), wgrad_ws() -- and at the entry of Trunk.forward /
        Trunk.backward (which use `partial` directly); one comparison per call when the stream has not changed."""
        if self.device.type != 'cuda':
            return
        cur = torch.cuda.current_stream(self.device)
        last = self.__dict__.get('_last_stream')
        if last is not None and last != cur:
            cur.wait_event(last.record_event())
        self._last_stream = cur

    def scratch(self, name, n):
        """a float32 scratch buffer of at least n elements, reused across calls of this thread on this stream"""
        if not name.endswith('_side'):      # (side-stream launches are ordered by their own events, _block_bwd)
            self.enter_stream()
        t = getattr(self, name, None)
        if t is None or t.numel() < n:
            t = torch.empty(n, dtype=torch.float32, device=self.device)
            setattr(self, name, t)
        return t

    def wgrad_ws(self, geo, cin, cout, side=False):
        n = int(_lib.load().mmlf_wgrad_workspace_floats(cin, cout, geo.B, geo.H, geo.W))
        if n < 0:
            raise RuntimeError(f'wgrad: unsupported channels {cin}->{cout}')
        name = 'wgrad_side' if side else 'wgrad'
        if not side:                 # (the side stream's launches are ordered by their own events, _block_bwd)
            self.enter_stream()
        if getattr(self, name) is None or getattr(self, name).numel() < n:
            setattr(self, name, torch.empty(n, dtype=torch.float32, device=self.device))
        return getattr(self, name)


# 'f16x3': 2-way f16 split of power-of-two-scaled operands, 3 MFMA passes (f32-equivalent accuracy, 5.3x the
# f32 MFMA rate); 'bf16x6': 3-way bf16 split, 6 passes (no operand scaling needed); 'f32': exact-f32 MFMA
CONV_MODE = os.environ.get('MMLF_CONV_MODE', 'f16x3')


def pack_filter(w, variant, dgrad):
    cout, cin = w.shape[0], w.shape[1]
    K, N = (cout, cin) if dgrad else (cin, cout)
    if CONV_MODE == 'f16x3':
        n = int(_lib.load().mmlf_packed_filter_h2_bytes(cs_of(K), N))
        if n < 0:
            raise RuntimeError(f'pack_filter: unsupported channels K={K} N={N}')
        out = torch.empty(n // 4, dtype=torch.float32, device=w.device)
        call('mmlf_pack_filter_h2', ptr(w), ptr(out), cout, cin, variant, int(dgrad), _lib.stream_ptr())
        return out
    if CONV_MODE == 'bf16x6':
        n = int(_lib.load().mmlf_packed_filter_split_bytes(cs_of(K), N))
        if n < 0:
            raise RuntimeError(f'pack_filter: unsupported channels K={K} N={N}')
        out = torch.empty(n // 4, dtype=torch.float32, device=w.device)
        call('mmlf_pack_filter_split', ptr(w), ptr(out), cout, cin, variant, int(dgrad), _lib.stream_ptr())
        return out
    # the kernel walks K in chunks of 8 over the channel stride of its input
    n = int(_lib.load().mmlf_packed_filter_floats(cs_of(K), N))
    if n < 0:
        raise RuntimeError(f'pack_filter: unsupported channels K={K} N={N}')
    out = torch.empty(n, dtype=torch.float32, device=w.device)
    call('mmlf_pack_filter', ptr(w), ptr(out), cout, cin, variant, int(dgrad), _lib.stream_ptr())
    # the kernel only fills ceil(K/8) chunks; cs_of(K)/8 == ceil(K/8)
    return out


PROFILE = None   # bench.py sets this to a list to time the 280-wide conv / weight-gradient launches with HIP events


def _amax_of(geo, t, cs):
    """The amax array (tensor and grid-row maxima of |t|) the f16-split kernels scale by.  Grid tensors made
    by Geometry.buf carry it (their producers maintain it); for any other tensor it is computed here."""
    a = getattr(t, 'absmax', None)
    if a is None:
        return geo.amax_of(t, cs)
    if CHECK_ABSMAX:       # test hook: the producers' running maxima must be the tensor's true maxima
        true = geo.amax_of(t, cs)
        c = geo.amax_canonical(a)
        exact = geo.P >= 32                 # smaller pitches: rows behind a wave's first get an upper bound
        bad = (c != true) if exact else (c < true)
        bad[0] = c[0] != true[0]
        if bool(bad.any()):
            k = int(bad.nonzero()[0])
            raise AssertionError(f'amax entry {k} holds {float(c[k])!r}, true max |x| is {float(true[k])!r}')
    return a


CHECK_ABSMAX = bool(os.environ.get('MMLF_CHECK_ABSMAX'))
# one launch packs every filter of a step / zeroes the slack of a block's buffers (0: per filter, per buffer)
BATCHED = os.environ.get('MMLF_BATCHED', '1') != '0'
# one BatchNorm-apply pass for the four streams' last blocks (whole rows of the concat buffer); 0: four slice passes
APPLY4 = os.environ.get('MMLF_APPLY4', '1') != '0'
# MMLF_OVERLAP_WGRAD (default 1 since round 6; 0 switches it off): conv1's weight gradient of the wide blocks runs on a side
# stream beside the BatchNorm-backward kernels of the block underneath (which only need the data gradient).  It pays since the
# weight gradient is down to 2 x 232 registers per SIMD and both BatchNorm-backward kernels fit the 48 left (round 5:
# +0.9...1.1 % on two boxes, gradients bit-identical, profiles/r05_overlap_modes.log; round 6's same-box A/B:
# profiles/r06_ab_overlap_wgrad.log).  The side-stream launch takes ~11.3 ms instead of 7.6 (it shares the CUs) while 4.7 ms of
# BatchNorm kernels hide behind it; +5.5 GiB stay alive (x and dy of one block, until the main stream has waited for the launch).
# The events around a side-stream launch do not time the kernel alone: bench.py's roofline_wgrad is taken on the main-stream
# launches (conv2's gradients: same kernel, same shape).  History of the rejected forms: EXPERIMENTS.md 4.9.
OVERLAP_WGRAD = os.environ.get('MMLF_OVERLAP_WGRAD', '1') not in ('', '0')


# MMLF_CHECK_EXTENTS=1 (debug; the f16-split launches): before every convolution / weight-gradient launch the host compares
# the audited END of what the launch may touch behind each pointer (mmlf_audit_conv_h2 / mmlf_audit_wgrad_h2: derived from the
# launch geometry) with the bytes the tensor behind that pointer really has, and raises instead of launching.  The product
# kernels' range-checked descriptors DROP a stray access (conv_device.h: mmlf_records_left), so a wrong extent would be a quietly
# wrong result; this is the product-build signal for it (the -DMMLF_BOUNDS_DEBUG build counts accesses on the GPU instead).
CHECK_EXTENTS = bool(os.environ.get('MMLF_CHECK_EXTENTS'))
EXTENT_CHECKS = 0         # launches checked so far (tests)


def _bytes_behind(t, off_floats=0):
    """bytes from a tensor's first element (+ an offset) to the end of its storage"""
    return t.untyped_storage().nbytes() - t.storage_offset() * t.element_size() - 4 * off_floats


def _check_extents(kind, ends, have):
    global EXTENT_CHECKS
    for name, (end, t, off) in have.items():
        if t is None:
            continue
        got = _bytes_behind(t, off)
        if end > got:
            raise RuntimeError(f'MMLF_CHECK_EXTENTS: {kind}: the launch may touch {end} bytes behind `{name}`, the tensor has {got}')
    EXTENT_CHECKS += 1


def _check_conv_extents(geo, x, cs_in, K, packed, bias, N, out, cs_out, n_store, out_off, out_shift, ref, cs_ref, ax, aout,
                        bn_partial, mask_out, mask_in):
    import ctypes
    e = (ctypes.c_int64 * 9)()
    call('mmlf_audit_conv_h2', cs_in, K, N, cs_out, n_store, out_shift, cs_ref, geo.B, geo.H, geo.W, e)
    _check_extents(f'conv {K}->{N} B={geo.B} {geo.H}x{geo.W} shift={out_shift}', e,
                   {'in': (e[0], x, 0), 'packed': (e[1], packed, 0), 'bias': (e[2], bias, 0), 'out': (e[3], out, out_off),
                    'ref': (e[4], ref, 0), 'in_amax': (e[5], ax, 0), 'out_amax': (e[6], aout, 0),
                    'bn_partial': (e[7], bn_partial, 0), 'mask_out': (e[8], mask_out, 0), 'mask_in': (e[8], mask_in, 0)})


def _check_wgrad_extents(geo, x, cs_in, cin, g, cs_g, cout, g_shift, gw, gb, workspace, ax, ag):
    import ctypes
    e = (ctypes.c_int64 * 7)()
    call('mmlf_audit_wgrad_h2', cs_in, cin, cs_g, cout, g_shift, geo.B, geo.H, geo.W, e)
    _check_extents(f'wgrad {cin}->{cout} B={geo.B} {geo.H}x{geo.W} g_shift={g_shift}', e,
                   {'in': (e[0], x, 0), 'g': (e[1], g, 0), 'gw': (e[2], gw, 0), 'gb': (e[3], gb, 0),
                    'workspace': (e[4], workspace, 0), 'in_amax': (e[5], ax, 0), 'g_amax': (e[6], ag, 0)})


THIN_MAX_N, THIN_MIN_K = 2, 64     # mmlf_conv2x2_thin: at most 2 output channels over at least 64 input channels


def wgrad(geo, x, cs_in, cin, g, cs_g, cout, g_shift, gw, gb, variant, workspace, side=False):
    """weight + bias gradient, accumulated into gw / gb (side: the launch goes to the side stream and must not share
    scratch with main-stream launches)"""
    if cout <= THIN_MAX_N and cin >= THIN_MIN_K and cs_in <= 512:
        # a matrix-vector product (the BASE / UPR head): plain float32 FMAs, bound by reading x once
        ws = _Workspace.get(x.device).scratch('thin_wgrad_side' if side else 'thin_wgrad', int(_lib.load().mmlf_conv2x2_wgrad_thin_workspace_floats(cin)))
        call('mmlf_conv2x2_wgrad_thin', ptr(x), cs_in, cin, ptr(g), cs_g, cout, g_shift, ptr(gw), ptr(gb), variant, 1,
             ptr(ws), geo.B, geo.H, geo.W, _lib.stream_ptr())
        return
    args = (ptr(x), cs_in, cin, ptr(g), cs_g, cout, g_shift, ptr(gw), ptr(gb), variant, 1, ptr(workspace),
            geo.B, geo.H, geo.W)
    prof = PROFILE is not None and cin >= 256 and cout >= 256
    if prof:      # events on the stream the launch goes to (the side stream for the overlapped ones)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    if CONV_MODE == 'f16x3':
        ax, ag = _amax_of(geo, x, cs_in), _amax_of(geo, g, cs_g)
        if CHECK_EXTENTS:
            _check_wgrad_extents(geo, x, cs_in, cin, g, cs_g, cout, g_shift, gw, gb, workspace, ax, ag)
        call('mmlf_conv2x2_wgrad_h2', *args, ptr(ax), ptr(ag), _lib.stream_ptr())
    else:
        call('mmlf_conv2x2_wgrad_split' if CONV_MODE == 'bf16x6' else 'mmlf_conv2x2_wgrad', *args, _lib.stream_ptr())
    if prof:
        e1.record()
        # algorithmic FLOPs: the convolution's valid output positions x Cout x 4 taps x Cin, 2 FLOP per MAC (the
        # gradient of a pad-1 convolution lives at grid offset 0 with extent (H+1, W+1), of a pad-0 one at (1, 1))
        vh, vw = (geo.H + 1, geo.W + 1) if g_shift == 0 else (geo.H, geo.W)
        nbytes = 4.0 * geo.B * (cin * (geo.H * geo.W if g_shift == 0 else (geo.H + 1) * (geo.W + 1)) + cout * vh * vw)
        PROFILE.append(('wgrad_side' if side else 'wgrad', 2.0 * geo.B * vh * vw * cout * 4 * cin, e0, e1, nbytes))


def conv(geo, x, cs_in, K, packed, bias, N, out, cs_out, out_shift, vh, vw, relu, ref=None, cs_ref=0,
         n_store=None, out_off=0, bn_partial=None, mask_out=None, mask_in=None, w_master=None, variant=0):
    """bn_partial (f16x3 only): a float64 buffer that receives per-workgroup sums of the output and its
    square per channel -- BatchNorm's training statistics without another pass over the output.
    mask_out / mask_in (f16x3 only): the ReLU mask of the output as bits (Geometry.relu_mask), written by the
    forward launch and read by the data gradient of the layer above instead of `ref`."""
    if (w_master is not None and N <= THIN_MAX_N and K >= THIN_MIN_K and cs_in <= 512 and ref is None and mask_in is None
            and mask_out is None and bn_partial is None and out_off == 0 and n_store in (None, cs_out)):
        # a matrix-vector product (the BASE / UPR head): straight from the OIHW master filter
        ws = _Workspace.get(x.device).scratch('thin_fwd', int(_lib.load().mmlf_conv2x2_thin_workspace_floats(geo.B, geo.H, geo.W)))
        call('mmlf_conv2x2_thin', ptr(x), cs_in, K, ptr(w_master), ptr(bias), N, ptr(out), cs_out, out_shift, vh, vw,
             geo.B, geo.H, geo.W, int(relu), variant, ptr(ws), ptr(getattr(out, 'absmax', None)), _lib.stream_ptr())
        return
    # bench.py's per-launch timing: the 280-wide launches (tag 'conv') and the 70 -> 70 stream-layer launches ('conv70')
    prof = PROFILE is not None and ((K >= 256 and N >= 256) or (K == N and 64 <= K < 128))
    if prof:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    args = (ptr(x), cs_in, K, ptr(packed), ptr(bias), N, ptr(out) + 4 * out_off, cs_out,
            cs_out if n_store is None else n_store, out_shift, vh, vw, geo.B, geo.H, geo.W, int(relu), ptr(ref), cs_ref)
    if CONV_MODE == 'f16x3':
        ax = _amax_of(geo, x, cs_in)
        if CHECK_EXTENTS:
            _check_conv_extents(geo, x, cs_in, K, packed, bias, N, out, cs_out, cs_out if n_store is None else n_store, out_off,
                                out_shift, ref, cs_ref, ax, getattr(out, 'absmax', None), bn_partial, mask_out, mask_in)
        call('mmlf_conv2x2_h2', *args, ptr(ax), ptr(getattr(out, 'absmax', None)), ptr(bn_partial), ptr(mask_out),
             ptr(mask_in), _lib.stream_ptr())
    else:
        call('mmlf_conv2x2_split' if CONV_MODE == 'bf16x6' else 'mmlf_conv2x2', *args, _lib.stream_ptr())
    if prof:
        e1.record()
        # algorithmic FLOPs of this launch: valid output positions x N x 4 taps x K, 2 FLOP per MAC; algorithmic BYTES: the
        # input's and the output's stored extents once each (a pad-1 convolution reads (H, W) and writes (H+1, W+1), a pad-0
        # one the other way round), float32, unpadded channels
        nbytes = 4.0 * geo.B * (K * (geo.H * geo.W if out_shift == 0 else (geo.H + 1) * (geo.W + 1)) + N * vh * vw)
        PROFILE.append(('conv' if K >= 256 else 'conv70', 2.0 * geo.B * vh * vw * N * 4 * K, e0, e1, nbytes))


class BlockSpec:
    def __init__(self, prefix, cin, cout, bn):
        self.prefix, self.cin, self.cout, self.bn = prefix, cin, cout, bn


class Trunk:
    """Native forward/backward of in_net_hv / in_net_id / out_net for the default flags
    (k=2, BatchNorm on, non-cross).  `params` maps state_dict keys to device tensors."""

    def __init__(self, chs, in_blocks, out_blocks, views, oc, momentum, eps=1e-5):
        self.chs, self.views, self.oc = chs, views, oc
        self.momentum, self.eps = float(momentum), float(eps)
        cin0 = views * 3
        self.streams = []
        for key, net, var in (('h', 'in_net_hv', VAR_TRANSPOSE), ('v', 'in_net_hv', VAR_IDENTITY),
                              ('i', 'in_net_id', VAR_TRANSPOSE_FLIPH), ('d', 'in_net_id', VAR_IDENTITY)):
            blocks = [BlockSpec(f'{net}.0', cin0, chs, True)]
            blocks += [BlockSpec(f'{net}.{k}', chs, chs, True) for k in range(1, in_blocks)]
            self.streams.append((key, var, blocks))
        c = 4 * chs
        self.out_blocks = [BlockSpec(f'out_net.{k}', c, c, True) for k in range(out_blocks - 1)]
        self.out_blocks.append(BlockSpec(f'out_net.{out_blocks - 1}', c, oc, False))
        if chs % 2 or cs_of(c) != c:
            raise ValueError('native trunk needs an even model_chs with 4*model_chs a multiple of 8')

    # ------------------------------------------------------------------ filters
    def _prepack(self, p, dev, with_dgrad):
        """every packed filter a step needs -- forward and, with_dgrad, data-gradient forms -- from ONE launch (f16 split
        only; the other modes pack per layer).  Keys: (parameter name, variant, dgrad)."""
        if CONV_MODE != 'f16x3' or dev.type != 'cuda' or not BATCHED:
            return {}
        items = []

        def add(name, var, dgrad):
            items.append(((name, var, dgrad), p[name], var, dgrad))

        def block(spec, var, first):
            thin = spec.cout <= THIN_MAX_N and spec.cin >= THIN_MIN_K      # the head's first conv runs from the master filter
            if not thin:
                add(f'{spec.prefix}.0.weight', var, False)
            add(f'{spec.prefix}.2.weight', var, False)
            if with_dgrad:
                add(f'{spec.prefix}.2.weight', var, True)
                if not first:
                    add(f'{spec.prefix}.0.weight', var, True)

        for _, var, blocks in self.streams:
            for k, spec in enumerate(blocks):
                block(spec, var, k == 0)
        for spec in self.out_blocks:
            block(spec, VAR_IDENTITY, False)
        return _Workspace.get(dev).packed_filters(items)

    # ------------------------------------------------------------------ forward
    def _block_fwd(self, geo, spec, var, x, cs_x, p, train, rec_list, out=None, cs_out=None, c_off=0, packs=None,
                   tracked=None, deferred=None):
        """x: grid tensor (extent H,W at (1,1)).  Returns the block output grid tensor.
        deferred (a list): the BatchNorm-apply + ReLU pass into `out` is NOT launched; (z, scale, shift) is appended
        and the caller applies all four streams' last blocks in one pass over the concat buffer (Trunk.forward)."""
        dev = x.device
        ws = _Workspace.get(dev)
        B, H, W, P = geo.B, geo.H, geo.W, geo.P
        cmid, cs_mid = spec.cout, cs_of(spec.cout)
        w1, b1 = p[f'{spec.prefix}.0.weight'], p[f'{spec.prefix}.0.bias']
        w2, b2 = p[f'{spec.prefix}.2.weight'], p[f'{spec.prefix}.2.bias']
        packs = packs or {}
        thin = cmid <= THIN_MAX_N and spec.cin >= THIN_MIN_K        # the head: matrix-vector kernels, y is tiny
        pk1 = packs.get((f'{spec.prefix}.0.weight', var, False))
        if pk1 is None and not thin:
            pk1 = pack_filter(w1, var, False)
        folded = spec.bn and not train and rec_list is None        # inference: BatchNorm folded into conv2
        new_out = spec.bn and out is None                          # the block output is a buffer of its own
        got = geo.bufs([cs_mid] * ((1 if folded else 2) + (1 if new_out else 0)), dev)   # one zeroing launch for all
        y = got[0]
        z = None if folded else got[1]
        if new_out:
            out, cs_out, c_off = got[-1], cs_mid, 0
        ymask = geo.relu_mask(dev) if (rec_list is not None and CONV_MODE == 'f16x3' and not thin) else None
        conv(geo, x, cs_x, spec.cin, pk1, b1, cmid, y, cs_mid, 0, H + 1, W + 1, True, mask_out=ymask, w_master=w1,
             variant=var)
        if spec.bn and not train and rec_list is None:   # rec_list is None when nothing is saved for backward
            # inference: BatchNorm(eval) is a per-channel affine map -> fold it into conv2 and fuse the ReLU
            C = spec.cout
            coef = torch.empty(2 * C, dtype=torch.float32, device=dev)
            call('mmlf_bn_coeffs_eval', ptr(p[f'{spec.prefix}.3.weight']), ptr(p[f'{spec.prefix}.3.bias']),
                 ptr(p[f'{spec.prefix}.3.running_mean']), ptr(p[f'{spec.prefix}.3.running_var']), self.eps,
                 ptr(coef), ptr(coef[C:]), C, _lib.stream_ptr())
            w2f, b2f = torch.empty_like(w2), torch.empty_like(b2)
            call('mmlf_fold_bn_eval', ptr(w2), ptr(b2), ptr(coef), ptr(coef[C:]), ptr(w2f), ptr(b2f), C, C,
                 _lib.stream_ptr())
            pk2 = pack_filter(w2f, var, False)
            n_store = cs_out if new_out else C
            conv(geo, y, cs_mid, cmid, pk2, b2f, cmid, out, cs_out, P + 1, H, W, True, n_store=n_store, out_off=c_off)
            return out, cs_out
        pk2 = packs.get((f'{spec.prefix}.2.weight', var, False))
        if pk2 is None:
            pk2 = pack_filter(w2, var, False)
        fused_stats = spec.bn and train and CONV_MODE == 'f16x3'      # statistics from the conv epilogue
        conv(geo, y, cs_mid, cmid, pk2, b2, cmid, z, cs_mid, P + 1, H, W, False,
             bn_partial=ws.partial if fused_stats else None)
        rec = {'spec': spec, 'var': var, 'x': x, 'cs_x': cs_x, 'y': y, 'z': z, 'ymask': ymask}
        if not spec.bn:
            if rec_list is not None:
                rec_list.append(rec)
            return z, cs_mid
        C = spec.cout
        coef = torch.empty(4 * C, dtype=torch.float32, device=dev)
        scale, shift, smean, sinv = coef[:C], coef[C:2 * C], coef[2 * C:3 * C], coef[3 * C:]
        g, bt = p[f'{spec.prefix}.3.weight'], p[f'{spec.prefix}.3.bias']
        rm, rv = p[f'{spec.prefix}.3.running_mean'], p[f'{spec.prefix}.3.running_var']
        if train:
            if fused_stats:
                nblk = int(_lib.load().mmlf_conv2x2_blocks(cmid, cmid, B, H, W))
                call('mmlf_bn_stats_finalize', ptr(ws.partial), nblk, C, ptr(g), ptr(bt), ptr(rm), ptr(rv),
                     self.momentum, self.eps, ptr(smean), ptr(sinv), ptr(scale), ptr(shift), B, H, W,
                     _lib.stream_ptr())
            else:
                call('mmlf_bn_stats_train', ptr(z), cs_mid, C, ptr(g), ptr(bt), ptr(rm), ptr(rv), self.momentum,
                     self.eps, ptr(smean), ptr(sinv), ptr(scale), ptr(shift), ptr(ws.partial), BN_BLOCKS, B, H, W,
                     _lib.stream_ptr())
            if tracked is None:
                p[f'{spec.prefix}.3.num_batches_tracked'].add_(1)
            else:
                tracked.append(p[f'{spec.prefix}.3.num_batches_tracked'])
        else:
            call('mmlf_bn_coeffs_eval', ptr(g), ptr(bt), ptr(rm), ptr(rv), self.eps, ptr(scale), ptr(shift), C,
                 _lib.stream_ptr())
            if rec_list is not None:
                # eval-mode BatchNorm under autograd (reference train/cli.py:227-230, --train_eval_mode): the
                # statistics are constants, so backward is dz = g * gamma * invstd with the RUNNING statistics
                smean.copy_(rm)
                sinv.copy_(torch.rsqrt(rv.double() + self.eps).float())
                rec['eval'] = True
        c_store = cs_out if new_out else C
        if deferred is not None:
            deferred.append((z, scale, shift))
        else:
            call('mmlf_bn_apply_relu', ptr(z), cs_mid, C, ptr(scale), ptr(shift), ptr(out), cs_out, c_off, c_store,
                 B, H, W, ptr(out.absmax), _lib.stream_ptr())
        rec.update(scale=scale, shift=shift, smean=smean, sinv=sinv)
        if rec_list is not None:
            rec_list.append(rec)
        return out, cs_out

    def forward(self, p, stacks, train, save, packed=None):
        """stacks: four (B, views, 3, H, W) contiguous float32 device tensors.
        packed (instead of stacks): (Geometry, [four grid tensors of channel stride cs_of(3 views), with their amax arrays]) --
        inputs some other kernel already wrote in the grid layout (the Ensamble's mmlf_shift_pack).
        Returns (output NCHW (B,oc,H,W), ctx or None)."""
        if packed is not None:
            geo, xs_in = packed
            B, H, W = geo.B, geo.H, geo.W
            dev = xs_in[0].device
            cin0 = self.views * 3
        else:
            h = stacks[0]
            B, n, c, H, W = h.shape
            dev = h.device
            geo = Geometry(B, H, W)
            cin0 = n * c
        _Workspace.get(dev).enter_stream()
        packs = self._prepack(p, dev, save)
        tracked = []                  # BatchNorm counters of this pass: ONE increment launch at its end
        tape = {'geo': geo, 'streams': [], 'out': [], 'packs': packs}
        if packed is not None:
            concat, xs = geo.buf(4 * self.chs, dev), list(xs_in)
        else:
            concat, *xs = geo.bufs([4 * self.chs] + [cs_of(cin0)] * 3, dev)
            xs.append(geo.buf(cs_of(cin0), dev))
        # the four streams' last BatchNorm-apply passes write quarter rows of the concat buffer: one pass for all four
        # (whole rows) when they are real passes (not folded into conv2) and the channel count allows it
        fold = not train and not save
        deferred = [] if (APPLY4 and not fold and self.chs % 2 == 0 and all(b[-1].bn for _, _, b in self.streams)) else None
        for s, (key, var, blocks) in enumerate(self.streams):
            x = xs[s]
            if packed is None:
                call('mmlf_pack_nchw', ptr(stacks[s]), cin0, ptr(x), cs_of(cin0), B, H, W, ptr(x.absmax), _lib.stream_ptr())
            cs_x = cs_of(cin0)
            recs = []
            for k, spec in enumerate(blocks):
                last = k == len(blocks) - 1
                x, cs_x = self._block_fwd(geo, spec, var, x, cs_x, p, train, recs if save else None,
                                          out=concat if last else None, cs_out=4 * self.chs, c_off=s * self.chs, packs=packs,
                                          tracked=tracked, deferred=deferred if last else None)
            tape['streams'].append(recs)
            if not save:
                del recs[:]
        if deferred:
            import ctypes
            arr = lambda k: (ctypes.c_void_p * 4)(*[ptr(d[k]) for d in deferred])
            call('mmlf_bn_apply_relu4', arr(0), cs_of(self.chs), self.chs, arr(1), arr(2), ptr(concat), 4 * self.chs,
                 B, H, W, ptr(concat.absmax), _lib.stream_ptr())
            del deferred[:]
        x, cs_x = concat, 4 * self.chs
        for spec in self.out_blocks:
            x, cs_x = self._block_fwd(geo, spec, VAR_IDENTITY, x, cs_x, p, train, tape['out'] if save else None, packs=packs,
                                      tracked=tracked)
            if not save:
                del tape['out'][:]
        if tracked:
            # the shared stream nets' counters appear twice: two forwards per pass, as in the reference
            # (feed_forward.py:222-235 calls in_net_hv for h and v) -- one entry per tensor with its count, since a
            # multi-tensor launch must not hold the same tensor twice
            counts = {}
            for t in tracked:
                counts.setdefault(t.data_ptr(), [t, 0])[1] += 1
            torch._foreach_add_([t for t, _ in counts.values()], [n for _, n in counts.values()])
        out = torch.empty((B, self.oc, H, W), dtype=torch.float32, device=dev)
        call('mmlf_unpack_nchw', ptr(x), cs_x, ptr(out), self.oc, B, H, W, _lib.stream_ptr())
        return out, (tape if save else None)

    # ------------------------------------------------------------------ backward
    def _block_bwd(self, geo, rec, p, grads, gy, cs_gy, c_off, need_dx, after_bn=None, overlap=False, packs=None):
        """gy: gradient w.r.t. the block output (grid, extent (H,W)).  Returns dX grid tensor.
        after_bn: called once this block's BatchNorm-backward kernels are enqueued.  overlap: run the
        first convolution's weight gradient on the side stream AFTER the data gradient is enqueued, so that
        it (matrix-core bound) runs beside the BatchNorm-backward kernels of the block underneath (HBM
        bound), which only need the data gradient; returns (dX, event, tensors to keep alive until the event) then."""
        spec, var = rec['spec'], rec['var']
        dev = gy.device
        ws = _Workspace.get(dev)
        B, H, W, P = geo.B, geo.H, geo.W, geo.P
        C, cs_mid = spec.cout, cs_of(spec.cout)
        x, cs_x, y, z = rec['x'], rec['cs_x'], rec['y'], rec['z']
        pre = spec.prefix
        sp = _lib.stream_ptr
        packs = packs or {}

        def packed(name, w):
            pk = packs.get((name, var, True))
            return pk if pk is not None else pack_filter(w, var, True)

        # this block's gradient buffers, one zeroing launch: dz (behind BatchNorm), dy, dx
        got = geo.bufs(([cs_mid] if spec.bn else []) + [cs_mid] + ([cs_x] if need_dx else []), dev)
        dy = got[1 if spec.bn else 0]
        dx = got[-1] if need_dx else None
        if spec.bn:
            coef = torch.empty(3 * C, dtype=torch.float32, device=dev)
            call('mmlf_bn_bwd_reduce', ptr(gy), cs_gy, c_off, ptr(z), cs_mid, C, ptr(rec['scale']), ptr(rec['shift']),
                 ptr(p[f'{pre}.3.weight']), ptr(rec['smean']), ptr(rec['sinv']), ptr(grads[f'{pre}.3.weight']),
                 ptr(grads[f'{pre}.3.bias']), 1, ptr(coef), ptr(ws.partial), BN_BLOCKS, B, H, W, sp())
            if rec.get('eval'):
                coef[C:].zero_()            # no batch-statistics terms: dz = k1 * g (dgamma / dbeta sums are the same)
            dz = got[0]
            call('mmlf_bn_bwd_apply', ptr(gy), cs_gy, c_off, ptr(z), cs_mid, C, ptr(rec['scale']), ptr(rec['shift']),
                 ptr(rec['smean']), ptr(coef), ptr(dz), cs_mid, B, H, W, ptr(dz.absmax), sp())
        else:
            assert c_off == 0 and cs_gy == cs_mid
            dz = gy
        if after_bn:
            after_bn()
        w1, w2 = p[f'{pre}.0.weight'], p[f'{pre}.2.weight']
        # conv2 (pad 0): weight/bias gradient, then data gradient fused with the ReLU mask of y
        wgrad(geo, y, cs_mid, C, dz, cs_mid, C, P + 1, grads[f'{pre}.2.weight'], grads[f'{pre}.2.bias'], var,
              ws.wgrad_ws(geo, C, C))
        pk = packed(f'{pre}.2.weight', w2)
        if rec.get('ymask') is not None and CONV_MODE == 'f16x3':
            conv(geo, dz, cs_mid, C, pk, None, C, dy, cs_mid, 0, H + 1, W + 1, False, mask_in=rec['ymask'])
        else:
            conv(geo, dz, cs_mid, C, pk, None, C, dy, cs_mid, 0, H + 1, W + 1, False, ref=y, cs_ref=cs_mid)
        del dz, got
        # conv1 (pad 1)
        if overlap and need_dx:
            pk = packed(f'{pre}.0.weight', w1)
            conv(geo, dy, cs_mid, C, pk, None, spec.cin, dx, cs_x, P + 1, H, W, False)
            main = torch.cuda.current_stream()
            ready = main.record_event()
            with torch.cuda.stream(ws.side):
                ws.side.wait_event(ready)
                wgrad(geo, x, cs_x, spec.cin, dy, cs_mid, C, 0, grads[f'{pre}.0.weight'], grads[f'{pre}.0.bias'],
                      var, ws.wgrad_ws(geo, spec.cin, C, side=True), side=True)
                done = ws.side.record_event()
            # x and dy are read by the side stream: the caller keeps them alive until the main stream has waited
            # for `done` (no record_stream: deferred reuse makes the caching allocator grow and stall)
            return dx, done, (x, dy)
        wgrad(geo, x, cs_x, spec.cin, dy, cs_mid, C, 0, grads[f'{pre}.0.weight'], grads[f'{pre}.0.bias'], var,
              ws.wgrad_ws(geo, spec.cin, C))
        if not need_dx:
            return None
        pk = packed(f'{pre}.0.weight', w1)
        conv(geo, dy, cs_mid, C, pk, None, spec.cin, dx, cs_x, P + 1, H, W, False)
        return dx

    def backward(self, p, tape, grad_output, grads, on_done=None):
        """grad_output: (B,oc,H,W) NCHW.  grads: dict name -> tensor, ACCUMULATED into
        (the caller zeroes them).  on_done(key) is called when every gradient of 'out_net.k' /
        'in_net_id' / 'in_net_hv' has been enqueued (gradient-bucket all-reduce hook)."""
        geo = tape['geo']
        dev = grad_output.device
        _Workspace.get(dev).enter_stream()
        B, H, W = geo.B, geo.H, geo.W
        cs = cs_of(self.oc)
        g = geo.buf(cs, dev)
        call('mmlf_pack_nchw', ptr(grad_output.contiguous()), self.oc, ptr(g), cs, B, H, W, ptr(g.absmax),
             _lib.stream_ptr())
        cs_g = cs
        recs = tape['out']
        main = torch.cuda.current_stream()
        pending = None                      # (event, prefix, tensors) of a weight gradient still running on the side stream

        def settle():
            nonlocal pending
            if pending is not None:
                main.wait_event(pending[0])
                if on_done:
                    on_done(pending[1])
                pending = None              # drops the tensors the side stream was reading

        while recs:
            rec = recs.pop()
            wide = OVERLAP_WGRAD and rec['spec'].cin >= 128
            res = self._block_bwd(geo, rec, p, grads, g, cs_g, 0, True, after_bn=settle, overlap=wide, packs=tape.get('packs'))
            settle()                        # (blocks without BatchNorm never called it)
            if wide:
                g, ev, keep = res
                pending = (ev, rec['spec'].prefix, keep)
            else:
                g = res
                if on_done:
                    on_done(rec['spec'].prefix)
            cs_g = rec['cs_x']
        # g is now the gradient w.r.t. the concat buffer (cs = 4*chs); streams read channel slices
        for s in reversed(range(4)):
            recs = tape['streams'][s]
            if s == 2:
                settle()                    # out_net.0's weight gradient ran beside the first stream's BatchNorm kernels
            gs, cs_s, off = g, cs_g, s * self.chs
            while recs:
                rec = recs.pop()
                gs = self._block_bwd(geo, rec, p, grads, gs, cs_s, off, need_dx=bool(recs), packs=tape.get('packs'))
                cs_s, off = rec['cs_x'], 0
            if on_done and s in (2, 0):      # shared stream nets: complete after the I (resp. H) stream
                on_done('in_net_id' if s == 2 else 'in_net_hv')
