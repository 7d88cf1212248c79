"""bf16 shadows of the fp32 master filters, refreshed by one kernel launch per optimizer step.

Under bf16 autocast every convolution used to cast its fp32 filter to bf16 on every forward (one 4 us launch each), and
the layers whose data gradient runs as a forward convolution rotated / transposed it with a second one: ~45 launches per
BiSeNet-R18 step that move 22 MB in total.  The modules of torchseg_amd.convwrw register their filters here instead;
`tsg_weight_shadow_refresh` rewrites every shadow in ONE launch.

Freshness: a shadow is valid for the version counter its parameter had when it was written.  Anything that changes the
parameter through torch (load_state_dict, an eager optimizer, init functions) bumps that counter, and the next `get`
refreshes.  FusedSGD updates parameters from its own kernel, which torch does not see — it therefore calls
`after_external_update()` at the end of every step (also inside a captured graph, where the refresh launch is captured with
the step).  NOT seen: in-place writes through `.data` (`p.data.mul_()`, `p.data.copy_()`, an EMA swap) — torch does not bump
the version counter for them.  Code that updates parameters that way must call
`torchseg_amd.shadow.after_external_update(device)` afterwards, or run with TSG_WEIGHT_SHADOW=0."""
import weakref

import numpy as np
import torch

from . import kernels as K

_CHUNK = 4096          # elements per block (kSgdChunk of csrc/sgd.hip)
# TSG_SHADOW_WF1_PASS=1|0 (default 1, round 6): the data-gradient fragment image written by blocks that walk the destination
_WF1_PASS = __import__("os").environ.get("TSG_SHADOW_WF1_PASS", "1") != "0"


class _Entry(object):
    __slots__ = ("ref", "wb", "wrt", "wf", "version")

    def __init__(self, param, want_rot):
        self.ref = weakref.ref(param)
        self.wb = torch.empty_like(param, dtype=torch.bfloat16)                       # same strides as the parameter
        self.wrt = None
        self.wf = {}                 # mode (0 forward / 1 data gradient) -> (fragment-order filter, tile width)
        if want_rot:
            self.add_rot(param)
        self.version = -1

    def add_rot(self, param):
        O, I = param.shape[0], param.shape[1]
        self.wrt = torch.empty((I, O, 3, 3), dtype=torch.bfloat16, device=param.device,
                               memory_format=torch.channels_last)


class _Bank(object):
    def __init__(self):
        self.entries = {}            # (data_ptr, shape, stride) of the parameter -> _Entry
        self._table = None           # (device table, block map, n_entries)
        self._subtables = {}         # parameter subset (sorted storage keys) -> the same, for refresh_subset

    def _prune(self):
        dead = [k for k, e in self.entries.items() if e.ref() is None]
        for k in dead:
            del self.entries[k]
        if dead:
            self._table = None
            self._subtables = {}

    @staticmethod
    def _key(param):
        # the storage, not the Python object: autograd hands a saved parameter back as an alias of the same storage
        return (param.data_ptr(), tuple(param.shape), tuple(param.stride()))

    def register(self, param, want_rot):
        e = self.entries.get(self._key(param))
        if e is not None and e.ref() is not None and e.wb.device == param.device:
            if want_rot and e.wrt is None:                # first asked for later than the plain cast: extend the entry
                self._check_rot(param)
                e.add_rot(param)
                e.version = -1
                self._table = None
                self._subtables = {}
            return e
        if param.dtype != torch.float32 or not param.is_cuda:
            raise K.L.TsgError("weight shadows are kept for fp32 parameters on the GPU")
        if want_rot:
            self._check_rot(param)
        if not (param.is_contiguous() or (param.dim() == 4 and param.is_contiguous(memory_format=torch.channels_last))):
            raise K.L.TsgError("weight shadows need a dense parameter")
        self._prune()
        e = self.entries[self._key(param)] = _Entry(param, want_rot)
        self._table = None
        self._subtables = {}
        return e

    @staticmethod
    def _check_rot(param):
        if not (param.dim() == 4 and tuple(param.shape[2:]) == (3, 3)
                and param.is_contiguous(memory_format=torch.channels_last)):
            raise K.L.TsgError("the rotated / fragment-order shadows need a channels_last [O, I, 3, 3] filter")

    def _build(self, device, ents=None):
        if ents is None:
            ents = [e for e in self.entries.values() if e.ref() is not None and e.wb.device == device]
        dt = np.dtype([("w", "<u8"), ("wb", "<u8"), ("wrt", "<u8"), ("wf0", "<u8"), ("wf1", "<u8"), ("n", "<i4"), ("O", "<i4"),
                       ("I", "<i4"), ("bn0", "<i4"), ("bn1", "<i4"), ("pad", "<i4")])
        assert dt.itemsize == K.provider().lib.tsg_weight_shadow_entry_bytes()
        tab = np.zeros(len(ents), dtype=dt)
        maps = []
        for i, e in enumerate(ents):
            p = e.ref()
            f0, f1 = e.wf.get(0), e.wf.get(1)
            # the data-gradient image in a pass of its own that walks the DESTINATION (csrc/sgd.hip weight_shadow_k, kind 1):
            # needs whole 16-byte vectors of 8 c_out and whole 32-lane groups of c_in
            own_pass = _WF1_PASS and f1 is not None and p.shape[0] % 16 == 0 and p.shape[1] % 32 == 0
            own_fwd = _WF1_PASS and f0 is not None and p.shape[1] % 16 == 0 and p.shape[0] % 32 == 0
            tab[i] = (p.data_ptr(), e.wb.data_ptr(), 0 if e.wrt is None else e.wrt.data_ptr(),
                      0 if f0 is None else f0[0].data_ptr(), 0 if f1 is None else f1[0].data_ptr(), p.numel(),
                      p.shape[0], p.shape[1] if p.dim() > 1 else 1, 0 if f0 is None else f0[1], 0 if f1 is None else f1[1],
                      (1 if own_pass else 0) | (2 if own_fwd else 0))
            nb = (p.numel() + _CHUNK - 1) // _CHUNK
            maps.append(np.stack([np.full(nb, i, dtype=np.int32), np.arange(nb, dtype=np.int32)], 1))
            if own_pass:
                maps.append(np.stack([np.full(nb, i, dtype=np.int32), np.arange(nb, dtype=np.int32) | (1 << 28)], 1))
            if own_fwd:
                maps.append(np.stack([np.full(nb, i, dtype=np.int32), np.arange(nb, dtype=np.int32) | (2 << 28)], 1))
        bmap = np.concatenate(maps, 0)
        table = torch.from_numpy(tab.view(np.uint8).copy()).to(device)
        return table, torch.from_numpy(np.ascontiguousarray(bmap)).to(device), ents

    def refresh_all(self, device):
        """One launch: every shadow on `device` from its parameter's current value."""
        self._prune()
        if not self.entries:
            return
        t = self._table
        if t is None or t[0].device != device or any(e.ref() is None or e.ref().data_ptr() != ptr for e, ptr in t[3]):
            table, bmap, ents = self._build(device)
            t = self._table = (table, bmap, ents, [(e, e.ref().data_ptr()) for e in ents])
        table, bmap, ents = t[0], t[1], t[2]
        if not ents:
            return
        L = K.L
        L.check(K.provider().lib.tsg_weight_shadow_refresh(table.data_ptr(), bmap.data_ptr(), bmap.shape[0],
                                                           L.stream_ptr(table)), "tsg_weight_shadow_refresh")
        for e in ents:
            p = e.ref()
            if p is not None:
                e.version = p._version

    def _subset_table(self, device, params):
        self._prune()
        ents = []
        for p in params:
            e = self.entries.get(self._key(p))
            if e is not None and e.ref() is not None and e.wb.device == device:
                ents.append(e)
        if not ents:
            return None
        key = tuple(sorted(id(e) for e in ents))
        t = self._subtables.get(key)
        if t is None or t[0].device != device or any(e.ref() is None or e.ref().data_ptr() != ptr for e, ptr in t[3]):
            table, bmap, ents = self._build(device, ents)
            t = self._subtables[key] = (table, bmap, ents, [(e, e.ref().data_ptr()) for e in ents])
        return t

    def prepare_subset(self, device, params):
        """Build the table refresh_subset(params) launches with (a host -> device copy: not allowed under stream capture)."""
        self._subset_table(device, params)

    def refresh_subset(self, device, params):
        """refresh_all for the shadows of `params` only (one launch; parameters without a shadow are skipped): an optimizer
        step taken in parts (FusedSGD.step(only=...)) rewrites what it changed and nothing else."""
        t = self._subset_table(device, params)
        if t is None:
            return
        table, bmap, ents = t[0], t[1], t[2]
        L = K.L
        L.check(K.provider().lib.tsg_weight_shadow_refresh(table.data_ptr(), bmap.data_ptr(), bmap.shape[0],
                                                           L.stream_ptr(table)), "tsg_weight_shadow_refresh")
        for e in ents:
            p = e.ref()
            if p is not None:
                e.version = p._version

    def get(self, param, want_rot=False):
        """(bf16 filter, rotated / transposed bf16 filter or None) of `param`, refreshed if the parameter changed."""
        e = self.register(param, want_rot)
        if e.version != param._version:
            self.refresh_all(param.device)
        return e.wb, e.wrt


    def get_gen(self, param, mode, bn):
        """The filter of `param` ([O, I, 3, 3] channels_last fp32 master) in the MFMA fragment order of csrc/conv3g.hip
        (what tsg_conv3x3_gen_prep_filter writes: mode 0 forward, mode 1 data gradient; `bn` = tile width), kept fresh by
        the same one launch per optimizer step as the plain bf16 casts — 32 preparation launches per BiSeNet step gone."""
        e = self.register(param, False)
        cur = e.wf.get(mode)
        if cur is None or cur[1] != bn:
            self._check_rot(param)
            O, I = param.shape[0], param.shape[1]
            Co, Ci = (I, O) if mode else (O, I)
            if Ci % 16 or Co % bn or bn not in (32, 64, 128):
                raise K.L.TsgError("fragment-order shadow: C_in %% 16 / C_out %% tile width (%d, %d, %d)" % (Ci, Co, bn))
            e.wf[mode] = (torch.empty(9 * O * I, dtype=torch.bfloat16, device=param.device), bn)
            e.version = -1
            self._table = None
            self._subtables = {}
        if e.version != param._version:
            self.refresh_all(param.device)
        return e.wf[mode][0]


bank = _Bank()


def after_external_update(device, params=None):
    """Called by an optimizer that writes parameters without torch noticing (FusedSGD): all shadows are rewritten, or
    those of `params` when the optimizer says which parameters it touched."""
    if bank.entries:
        if params is None:
            bank.refresh_all(device)
        else:
            bank.refresh_subset(device, params)
