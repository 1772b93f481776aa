"""Host-side mirror of the reference's `ActiveInferenceModel` (/root/reference/src/torchmodel.py:149-393)
for the EFE hot path, backed by the HIP engine through the C ABI of include/efe_engine.h.

Same attribute names, method names, argument meaning and return tuples as the reference, so
`src/mcts.py` / `src/util.py`-style callers are drop-in.  PyTorch is used only for device memory
and streams; every network evaluation and every EFE reduction runs in libefe_mi355x.so, reached through the
`torch.ops.efe.*` custom ops (csrc/torch_ops.cpp -> libefe_torch_ops.so) that are registered over the C ABI; the
ctypes binding of the same ABI (_lib.py) serves context management, the tree / environment kernels and the ABI tests.
There is no CPU fallback: constructing a model without a HIP device raises.

Noise: the reference draws MC-dropout masks and normals from torch's unseeded global generator; here
every draw is a pure function of (seed, stage, pass, sample, global row, element) -- csrc/philox.h.
`stage` is a per-model call counter (one `calculate_G` call = one stage), overridable per call with
`stage=` for reproducible parity tests; `row_offset` is the global index of local row 0 (multi-GPU).
"""
import ctypes as C
import os
import sys
import pickle

import numpy as np
import torch

from . import _lib

PASS_T1, PASS_D1, PASS_E1, PASS_T2, PASS_D2A, PASS_D2B, PASS_ROOT, PASS_HABIT, PASS_SIM = range(9)

def layer_shapes(pi_dim=4, channels=1, resolution=64):
    """state_dict tensor shapes per sub-model (reference key names, torchmodel.py:13-128).  (pi 4, 1 x 64 x 64) is the reference's
    Dynamic-dSprites model (with the 576-input repair of the first encoder Linear); other geometries are
    build-defined: four Conv2d(k3, s2) leave h4 x h4 x 64 features, the decoder starts from 64 x res/4 x res/4."""
    h = resolution
    for _ in range(4):
        h = (h - 3) // 2 + 1
    base = resolution // 2 if resolution == 32 else resolution // 4      # resolution 32: the reference's own variant (last_strides = 1, torchmodel.py:79-80)
    return {
        'top': {'qpi_net.0': (128, 10), 'qpi_net.2': (128, 128), 'qpi_net.4': (pi_dim, 128)},
        'mid': {'ps_net.0': (512, pi_dim + 10), 'ps_net.3': (512, 512), 'ps_net.6': (512, 512), 'ps_net.9': (20, 512)},
        'down': {'qs_net.0': (32, channels, 3, 3), 'qs_net.2': (32, 32, 3, 3), 'qs_net.4': (64, 32, 3, 3), 'qs_net.6': (64, 64, 3, 3),
                 'qs_net.9': (256, 64 * h * h), 'qs_net.12': (256, 256), 'qs_net.15': (256, 256), 'qs_net.18': (20, 256),
                 'po_net.0': (256, 10), 'po_net.3': (256, 256), 'po_net.6': (256, 256), 'po_net.9': (64 * base * base, 256),
                 'po_net.13': (64, 64, 3, 3), 'po_net.15': (64, 64, 3, 3), 'po_net.17': (64, 32, 3, 3), 'po_net.19': (32, channels, 3, 3)},
    }


_SHAPES = layer_shapes()
_CONVT = ('po_net.13', 'po_net.15', 'po_net.17', 'po_net.19')


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


class _Engine:
    """Owns one efe_ctx on one device."""

    def __init__(self, device_index, cfg=(10, 4, 1, 64)):
        self.lib = _lib.load()
        self.ops = _lib.load_ops()
        if not torch.cuda.is_available():
            raise RuntimeError('deep-active-inference-mc_amd needs a HIP device (MI355X); there is no CPU fallback')
        self.device = torch.device('cuda', device_index)
        self.ctx = C.c_void_p()
        rc = self.lib.efe_create_cfg(C.byref(self.ctx), device_index, *[int(v) for v in cfg])
        if rc == 7:
            raise ValueError(f'unsupported model geometry (s_dim, pi_dim, colour_channels, resolution) = {tuple(cfg)}: the engine takes '
                             's_dim 10, pi_dim 2..6, 1..3 channels, resolution a multiple of 4 in [32, 128]')
        if rc != 0:
            raise RuntimeError(f'efe_create failed with code {rc}')
        self.h = int(self.ctx.value)              # context handle as the torch.ops.efe ops take it
        # development hook: EFE_ENGINE_OPTS="name=value,..." applies efe_set_option to every context (kernel A/B experiments)
        for kv in filter(None, os.environ.get('EFE_ENGINE_OPTS', '').split(',')):
            k, v = kv.split('=')
            self.check(self.lib.efe_set_option(self.ctx, k.encode(), int(v)))

    def check(self, rc):
        if rc != 0:
            raise RuntimeError('efe engine: ' + self.lib.efe_last_error(self.ctx).decode())

    def stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def tensor(self, x, shape=None):
        t = torch.as_tensor(x)
        t = t.to(device=self.device, dtype=torch.float32).contiguous()
        if shape is not None:
            t = t.reshape(shape)
        return t

    def empty(self, *shape):
        return torch.empty(*shape, dtype=torch.float32, device=self.device)

    def __del__(self):
        try:
            # at interpreter shutdown the HIP runtime / torch allocator may already be half torn down: the process is about to
            # release everything anyway (one suite run in ~10 ended with a fatal signal after "N passed" before this guard)
            if sys.is_finalizing():
                return
            if self.ctx:
                self.lib.efe_destroy(self.ctx)
                self.ctx = C.c_void_p()
        except Exception:
            pass


class _Module:
    """state_dict holder with the reference's key names (torchmodel.py:167-177)."""

    def __init__(self, owner, part):
        self._owner = owner
        self._part = part
        self._sd = {}

    def state_dict(self):
        return {k: v.clone() for k, v in self._sd.items()}

    def load_state_dict(self, sd):
        want = self._owner._shapes[self._part]
        new = {}
        for name, shape in want.items():
            for suffix in ('weight', 'bias'):
                key = f'{name}.{suffix}'
                if key not in sd:
                    raise KeyError(f'missing key {key} in {self._part} state_dict')
                t = torch.as_tensor(sd[key]).detach().to('cpu', torch.float32).contiguous()
                if suffix == 'weight':
                    exp = shape
                else:
                    exp = (shape[1],) if name in _CONVT else (shape[0],)
                if tuple(t.shape) != tuple(exp):
                    hint = ''
                    if key == 'qs_net.9.weight' and tuple(t.shape) in ((256, 256), (256, 64)):
                        hint = (' (this is the shipped port defect torchmodel.py:94: the encoder emits 576 features; '
                                'such a checkpoint cannot run the reference either)')
                    raise ValueError(f'{self._part}.{key}: shape {tuple(t.shape)} != {tuple(exp)}{hint}')
                new[key] = t
        self._sd = new
        self._owner._weights_dirty = True
        self._owner._weights_version = getattr(self._owner, '_weights_version', 0) + 1

    def parameters(self):
        return list(self._sd.values())

    def to(self, *a, **k):
        return self


class ModelTop(_Module):
    """torchmodel.py:10-31"""

    def __init__(self, owner):
        super().__init__(owner, 'top')
        self.s_dim, self.pi_dim = owner.s_dim, owner.pi_dim

    def encode_s(self, s0):
        m = self._owner
        e = m._ready()
        s0 = e.tensor(s0, (-1, m.s_dim))
        return e.ops.habit(e.h, s0)


class ModelMid(_Module):
    """torchmodel.py:34-66"""

    def __init__(self, owner):
        super().__init__(owner, 'mid')
        self.s_dim, self.pi_dim = owner.s_dim, owner.pi_dim

    def reparameterize(self, mean, logvar, **kw):
        """torchmodel.py:54-56; the normals come from the model's Philox stream (one stage per call)"""
        return self._owner._reparameterize(mean, logvar, **kw)

    def transition_with_sample(self, pi, s0, stage=None, pass_=PASS_T1, sample=0, eps=None, row_offset=None):
        m = self._owner
        e = m._ready()
        pi = e.tensor(pi, (-1, m.pi_dim)); s0 = e.tensor(s0, (-1, m.s_dim))
        M = s0.shape[0]
        nz = m._noise(stage, pass_, sample, row_offset)
        if eps is None and m.eps_source is not None:
            eps = m._src_eps(M, 10, pass_, sample, nz.stage, row_offset)
        eps_t = e.tensor(eps, (M, 10)) if eps is not None else None
        return e.ops.transition(e.h, pi, s0, m._seed64(), nz.stage, pass_, sample, nz.row_offset, eps_t)

    def transition(self, pi, s0, **kw):
        _, mean, logvar = self.transition_with_sample(pi, s0, **kw)
        return mean, logvar


class ModelDown(_Module):
    """torchmodel.py:69-146 (resolution 64, 1 colour channel; the first encoder Linear takes the 576
    features the conv trunk actually emits -- SURVEY appendix C)."""

    def __init__(self, owner):
        super().__init__(owner, 'down')
        self.s_dim, self.pi_dim = owner.s_dim, owner.pi_dim
        self.colour_channels, self.resolution = owner.colour_channels, owner.resolution

    def reparameterize(self, mean, logvar, **kw):
        """torchmodel.py:130-132; the normals come from the model's Philox stream (one stage per call)"""
        return self._owner._reparameterize(mean, logvar, **kw)

    def encoder_with_sample(self, o, stage=None, pass_=PASS_E1, sample=0, eps=None, row_offset=None, _want_s=True):
        m = self._owner
        e = m._ready()
        o = e.tensor(o, (-1, m.colour_channels, m.resolution, m.resolution))
        M = o.shape[0]
        nz = m._noise(stage, pass_, sample, row_offset)
        if eps is None and _want_s and m.eps_source is not None:
            eps = m._src_eps(M, 10, pass_, sample, nz.stage, row_offset)
        eps_t = e.tensor(eps, (M, 10)) if eps is not None else None
        s, mean, logvar = e.ops.encoder(e.h, o, m._seed64(), nz.stage, pass_, sample, nz.row_offset, eps_t, bool(_want_s))
        return (s if _want_s else None), mean, logvar

    def encoder(self, o, **kw):
        kw.setdefault('pass_', PASS_ROOT)
        _, mean, logvar = self.encoder_with_sample(o, _want_s=False, **kw)
        return mean, logvar

    def decoder(self, s, stage=None, pass_=PASS_D1, sample=0, row_offset=None):
        m = self._owner
        e = m._ready()
        s = e.tensor(s, (-1, m.s_dim))
        M = s.shape[0]
        nz = m._noise(stage, pass_, sample, row_offset)
        return e.ops.decoder(e.h, s, m._seed64(), nz.stage, pass_, sample, nz.row_offset)


class Rows:
    """The row set of ONE calculate_G / calculate_G_mean / simulate_batch call (efe_rows, include/efe_engine.h): the rows of the call are
    entries of `rows_per_entry` consecutive rows (the pi_dim action rows of an episode; simulate_batch: one episode = one entry).
      mask : uint8 device tensor indexed by entry ID, read when the kernels run -- dead entries are skipped, their outputs unspecified
      ids  : int32 device tensor, entry slot -> entry ID: a COMPACTED call (only the live episodes) that still draws the noise of the
             episodes it holds; `ids_host` (the same indices as a host sequence) is needed only with injected noise (eps_source)
      n_total : entries of the un-compacted batch (length of `mask`, exclusive bound of every id); 0 = not stated.  The engine rejects a
             call with more entries, and with set_option('check_rows', 1) range-checks the ids on the host before the launch."""

    def __init__(self, mask=None, ids=None, rows_per_entry=1, ids_host=None, n_total=0):
        for t, dt, nm in ((mask, torch.uint8, 'mask'), (ids, torch.int32, 'ids')):
            if t is not None and (t.dtype != dt or not t.is_cuda or not t.is_contiguous()):
                raise ValueError(f'Rows.{nm}: a contiguous {dt} tensor on the model device is required')
        self.mask, self.ids, self.rows_per_entry, self.n_total = mask, ids, int(rows_per_entry), int(n_total)
        # (kept by reference: the planner builds one Rows per compaction and reuses it for every call of the iterations that follow)
        self.ids_host = ids_host

    def host_rows(self, div):
        """global-within-rank row index of every row of the call, for injected noise"""
        if self.ids is None:
            return None
        ids = self.ids_host if self.ids_host is not None else self.ids.cpu().tolist()
        return np.array([i * div + k for i in ids for k in range(div)], dtype=np.int64)


class ActiveInferenceModel:
    """Drop-in for the reference class on the EFE hot path.  Extra keyword-only arguments
    (`device`, `seed`, `row_offset`) configure the engine; everything else follows torchmodel.py:150."""

    def __init__(self, s_dim, pi_dim, gamma, beta_s, beta_o, colour_channels=1, resolution=64, *, device=None, seed=0,
                 row_offset=0, init_weights=True):
        # (10, 4, 1, 64) is the reference's Dynamic-dSprites model (fused kernels, parity pinned).  Other geometries -- BASELINE
        # configs[4]: pi 3, 3 x 84 x 84 -- are build-defined and parity-unpinned: the reference rejects the resolution
        # (torchmodel.py:77-82) and its Animal-AI reward is undefined (torchmodel.py:214); see layer_shapes() / efe_create_cfg.
        if device is None:
            idx = torch.cuda.current_device() if torch.cuda.is_available() else 0
        else:
            idx = torch.device(device).index or 0
        self.colour_channels, self.resolution = int(colour_channels), int(resolution)
        self._engine = _Engine(idx, (s_dim, pi_dim, colour_channels, resolution))
        self.device = self._engine.device
        self.s_dim, self.pi_dim = s_dim, pi_dim
        self._shapes = layer_shapes(pi_dim, self.colour_channels, self.resolution)
        self.parity_pinned = (pi_dim, self.colour_channels, self.resolution) == (4, 1, 64)
        self.seed = int(seed)
        self.row_offset = int(row_offset)
        self._stage = 0
        self._weights_dirty = True
        self.precision = torch.float32
        # Optional injected-noise mode (parity tests): callables with the signatures of the Philox mirror,
        #   eps_source(seed, rows, n, pass_, sample, stage, row_offset) -> float32 [rows, n] normals
        #   u_source(seed, rows, pass_, sample, stage, row_offset)      -> float32 [rows] uniforms in (0,1)
        # When set, every call that was not given explicit `eps` builds them from the source instead of using the device
        # generator (same Philox stream, but Box-Muller evaluated by the source's libm: bit-equal normals on both sides).
        self.eps_source = None
        self.u_source = None
        self.model_top = ModelTop(self)
        self.model_mid = ModelMid(self)
        self.model_down = ModelDown(self)
        self.beta_s = torch.tensor(beta_s, device=self.device)
        self.gamma = torch.tensor(gamma, device=self.device)
        self.beta_o = torch.tensor(beta_o, device=self.device)
        self.pi_one_hot = torch.eye(4, device=self.device)
        self.pi_one_hot_3 = torch.eye(3, device=self.device)
        if init_weights:
            self._init_random(seed)

    # ---- weights ----------------------------------------------------------------------------------
    def _init_random(self, seed):
        """He-uniform-like random init (the reference comments 'He Uniform', torchmodel.py:14)."""
        g = torch.Generator().manual_seed(int(seed) & 0x7FFFFFFF)
        for part, mod in (('top', self.model_top), ('mid', self.model_mid), ('down', self.model_down)):
            sd = {}
            for name, shape in self._shapes[part].items():
                if len(shape) == 2:
                    fan = shape[1]
                elif name in _CONVT:
                    fan = shape[0] * 9 / (4 if name in ('po_net.15', 'po_net.17') else 1)
                else:
                    fan = shape[1] * 9
                bound = (1.0 if part == 'mid' else 1.15) * (3.0 / fan) ** 0.5
                sd[f'{name}.weight'] = (torch.rand(shape, generator=g) * 2 - 1) * bound
                nb = shape[1] if name in _CONVT else shape[0]
                sd[f'{name}.bias'] = (torch.rand(nb, generator=g) * 2 - 1) * 0.1
            if part == 'mid':
                sd['ps_net.9.weight'] *= 0.3       # keep the depth recursion of imagined states bounded
                sd['ps_net.9.bias'][10:] -= 2.0
            if part == 'down':
                sd['qs_net.18.bias'][10:] -= 2.0
            mod.load_state_dict(sd)

    def load_state_dicts(self, top, mid, down):
        self.model_top.load_state_dict(top)
        self.model_mid.load_state_dict(mid)
        self.model_down.load_state_dict(down)

    def load_flat_weights(self, weights):
        """weights: dict '<top|mid|down>.<state_dict key>' -> array (oracle/synth.py naming)."""
        parts = {'top': {}, 'mid': {}, 'down': {}}
        for k, v in weights.items():
            part, key = k.split('.', 1)
            parts[part][key] = torch.as_tensor(np.asarray(v))
        self.load_state_dicts(parts['top'], parts['mid'], parts['down'])

    def _ready(self):
        e = self._engine
        if self._weights_dirty:
            for part, mod in (('top', self.model_top), ('mid', self.model_mid), ('down', self.model_down)):
                for key, t in mod._sd.items():
                    shape = (C.c_int64 * t.dim())(*t.shape)
                    e.check(e.lib.efe_set_weight(e.ctx, f'{part}.{key}'.encode(), C.c_void_p(t.data_ptr()), shape, t.dim()))
            e.check(e.lib.efe_commit_weights(e.ctx))
            self._weights_dirty = False
        return e

    def replica(self):
        """a second engine context on the same device with the same geometry, seed and weights: lets two engine calls run
        concurrently on two HIP streams (one context has one scratch arena and orders its calls)"""
        r = ActiveInferenceModel(self.s_dim, self.pi_dim, float(self.gamma), float(self.beta_s), float(self.beta_o),
                                 colour_channels=self.colour_channels, resolution=self.resolution, device=self.device, seed=self.seed,
                                 row_offset=self.row_offset, init_weights=False)
        r.load_state_dicts(self.model_top._sd, self.model_mid._sd, self.model_down._sd)
        r.eps_source, r.u_source = self.eps_source, self.u_source
        for name, value in getattr(self, '_opts', {}).items():        # engine options are per context: the replica computes what this model computes
            r.set_option(name, value)
        return r

    def cached_replica(self):
        """the replica the lock-step planner runs its simulations on, created once per weight version (a replica packs and uploads
        all weights: ~40 ms -- per planning decision it was 12 % of configs[2]) and refreshed with this model's noise settings"""
        ver = getattr(self, '_weights_version', 0)
        r = getattr(self, '_replica', None)
        if r is None or self._replica_version != ver:
            r = self.replica()
            self._replica, self._replica_version = r, ver
        r.seed, r.row_offset = self.seed, self.row_offset
        r.eps_source, r.u_source = self.eps_source, self.u_source
        mine, theirs = getattr(self, '_opts', {}), getattr(r, '_opts', {})
        for name, value in mine.items():
            if theirs.get(name) != value:
                r.set_option(name, value)
        return r

    def reserve(self, rows, steps, samples):
        """pre-size the engine's scratch arena for calculate_G_repeated(rows, steps, samples): later calls never hipMalloc"""
        e = self._ready()
        need = int(e.lib.efe_rollout_scratch_bytes(e.ctx, int(rows), int(steps), int(samples)))
        e.check(e.lib.efe_reserve(e.ctx, need))
        return need

    def engine_device(self):
        """-> (HIP device index, PCI bus id) of the engine context (efe_get_device): the GPU the kernels of this model run on"""
        e = self._engine
        dev = C.c_int(-1)
        buf = C.create_string_buffer(64)
        e.check(e.lib.efe_get_device(e.ctx, C.byref(dev), buf, 64))
        return dev.value, buf.value.decode()

    def arena_stats(self):
        """-> dict(capacity_bytes, high_water_bytes, grow_count)"""
        e = self._engine
        cap, hw, gr = C.c_int64(), C.c_int64(), C.c_int64()
        e.check(e.lib.efe_arena_stats(e.ctx, C.byref(cap), C.byref(hw), C.byref(gr)))
        return {'capacity_bytes': cap.value, 'high_water_bytes': hw.value, 'grow_count': gr.value}

    def set_option(self, name, value):
        """efe_set_option: launch-group sizes (dec_chunk, enc_chunk, dec_chunk_g, dec_budget_g), reward_upstream_intent (0 / 1: the reward
        target the upstream NHWC code means instead of the shipped port's NCHW broadcast), A/B switches that leave the
        results unchanged (generic path: fuse_final_g, ct_fuse12, enc_tiled; sim_split, mid_unfused, head_unfused), the mfma_bf16x3 /
        mfma_f16x2 experiments (one split mode at a time: setting one replaces the other), poison / trace / arena_align / check_rows (development) -- the list with defaults: include/efe_engine.h"""
        e = self._engine
        e.check(e.lib.efe_set_option(e.ctx, name.encode(), int(value)))
        self._opts = dict(getattr(self, '_opts', {}), **{name: int(value)})
        if name in ('mfma_bf16x3', 'mfma_f16x2'):          # ONE split mode in the engine: setting either option replaces (or, with 0, clears) the other
            other = 'mfma_f16x2' if name == 'mfma_bf16x3' else 'mfma_bf16x3'
            self._opts.pop(other, None)
            self._opts.pop(name, None)
            self._opts[name] = int(value)                  # (re-inserted last: a replica replays the options in this order)
        r = getattr(self, '_replica', None)
        if r is not None:                   # the planner's simulation context follows (one tree must not mix two reward definitions)
            r.set_option(name, value)

    def save_weights(self, folder_chp):
        """torchmodel.py:167-171"""
        torch.save(self.model_down.state_dict(), f'{folder_chp}/checkpoint_down.pth')
        torch.save(self.model_top.state_dict(), f'{folder_chp}/checkpoint_top.pth')
        torch.save(self.model_mid.state_dict(), f'{folder_chp}/checkpoint_mid.pth')

    def load_weights(self, folder_chp):
        """torchmodel.py:173-177"""
        self.model_down.load_state_dict(torch.load(f'{folder_chp}/checkpoint_down.pth', map_location='cpu'))
        self.model_top.load_state_dict(torch.load(f'{folder_chp}/checkpoint_top.pth', map_location='cpu'))
        self.model_mid.load_state_dict(torch.load(f'{folder_chp}/checkpoint_mid.pth', map_location='cpu'))

    def save_all(self, folder_chp, stats, script_file='', optimizers={}):
        """torchmodel.py:179-189 (weights + stats; optimiser state is training-only and out of scope)."""
        self.save_weights(folder_chp)
        with open(f'{folder_chp}/stats.pkl', 'wb') as ff:
            pickle.dump(stats, ff)

    def load_all(self, folder_chp):
        """torchmodel.py:191-208"""
        self.load_weights(folder_chp)
        stats = {}
        if os.path.exists(f'{folder_chp}/stats.pkl'):
            with open(f'{folder_chp}/stats.pkl', 'rb') as ff:
                stats = pickle.load(ff)
        if stats.get('var_beta_s'):
            self.beta_s = torch.tensor(stats['var_beta_s'][-1], device=self.device)
        if stats.get('var_gamma'):
            self.gamma = torch.tensor(stats['var_gamma'][-1], device=self.device)
        if stats.get('var_beta_o'):
            self.beta_o = torch.tensor(stats['var_beta_o'][-1], device=self.device)
        return stats, {}

    def to(self, *a, **k):
        return self

    def parameters(self):
        return self.model_top.parameters() + self.model_mid.parameters() + self.model_down.parameters()

    # ---- noise bookkeeping --------------------------------------------------------------------------
    def _take_stage(self, stage, n=1):
        if stage is None:
            stage = self._stage
            self._stage += n
        return int(stage)

    def _src_eps(self, rows, n, pass_, sample, stage, row_offset):
        ro = self.row_offset if row_offset is None else int(row_offset)
        return np.asarray(self.eps_source(self.seed, rows, n, pass_, sample, int(stage), ro), dtype=np.float32)

    def _src_eps_calcG(self, M, S, stage, row_offset):
        """[3S, M, 10]: T1_0..T1_{S-1}, T2_*, D2B_* (layout of efe_calculate_g)"""
        return np.stack([self._src_eps(M, 10, pas, i, stage, row_offset) for pas in (PASS_T1, PASS_T2, PASS_D2B) for i in range(S)], 0)

    def _seed64(self):
        """the 64-bit noise seed as the signed integer a torch op schema carries"""
        s = self.seed & 0xFFFFFFFFFFFFFFFF
        return s - (1 << 64) if s >= (1 << 63) else s

    def _noise(self, stage, pass_, sample, row_offset=None):
        return _lib.EfeNoise(self.seed, self._take_stage(stage), pass_, sample,
                             self.row_offset if row_offset is None else int(row_offset))

    # ---- hot path -----------------------------------------------------------------------------------
    def check_reward(self, o):
        """torchmodel.py:210-212 on an arbitrary image batch [M,1,64,64] -> [M] (the rollout path computes the same
        expression inside the fused decoder epilogue)"""
        e = self._ready()
        return e.ops.check_reward(e.h, e.tensor(o, (-1, self.colour_channels, self.resolution, self.resolution)))

    def _reparameterize(self, mean, logvar, stage=None, pass_=PASS_ROOT, sample=0, eps=None, row_offset=None):
        e = self._ready()
        mean = e.tensor(mean); logvar = e.tensor(logvar)
        M, n = mean.shape[0], mean.shape[1]
        nz = self._noise(stage, pass_, sample, row_offset)
        if eps is None and self.eps_source is not None:
            eps = self._src_eps(M, n, pass_, sample, nz.stage, row_offset)
        eps_t = e.tensor(eps, (M, n)) if eps is not None else None
        return e.ops.reparameterize(e.h, mean, logvar, self._seed64(), nz.stage, pass_, sample, nz.row_offset, eps_t)

    def imagine_future_from_o(self, o0, pi):
        """torchmodel.py:216-220"""
        s0, _, _ = self.model_down.encoder_with_sample(o0, pass_=PASS_ROOT)
        ps1, _, _ = self.model_mid.transition_with_sample(pi, s0)
        return self.model_down.decoder(ps1)

    def habitual_net(self, o):
        """torchmodel.py:222-225"""
        qs_mean, _ = self.model_down.encoder(o)
        _, Qpi, _ = self.model_top.encode_s(qs_mean)
        return Qpi

    def calculate_G(self, s0, pi0, samples=10, *, stage=None, eps=None, row_offset=None, _mean_mode=False, _parts=None, rows=None):
        """torchmodel.py:270-300 -> (G, [term0, term1, term2], ps1, ps1_mean, po1); rows: optional Rows (liveness mask / compacted batch)"""
        e = self._ready()
        s0 = e.tensor(s0, (-1, self.s_dim)); pi0 = e.tensor(pi0, (-1, self.pi_dim))
        M = s0.shape[0]
        nz = self._noise(stage, 0, 0, row_offset)
        S = 1 if _mean_mode else int(samples)
        if S < 1:
            raise RuntimeError('efe engine: samples must be >= 1')
        if eps is None and self.eps_source is not None:
            src_stage = nz.stage
            hr = rows.host_rows(rows.rows_per_entry) if rows is not None else None
            if hr is None:
                eps = self._src_eps_calcG(M, S, src_stage, row_offset)
            else:                                               # a compacted call: the normals of the rows it holds
                eps = self._src_eps_calcG(int(hr.max()) + 1, S, src_stage, row_offset)[:, hr]
        eps_t = e.tensor(eps, (3 * S, M, 10)) if eps is not None else None
        rm, ri, rpe, rnt = (rows.mask, rows.ids, rows.rows_per_entry, rows.n_total) if rows is not None else (None, None, 1, 0)
        G, terms, ps1, ps1_mean, po1, parts = e.ops.calculate_g(e.h, s0, pi0, S, bool(_mean_mode), self._seed64(), nz.stage,
                                                                nz.row_offset, eps_t, rm, ri, rpe, rnt)
        if _parts is not None:
            _parts.append(parts)
        if _mean_mode:
            return G, [terms[0], terms[1], terms[2]], ps1_mean, po1
        return G, [terms[0], terms[1], terms[2]], ps1, ps1_mean, po1

    def calculate_G_mean(self, s0, pi0, *, stage=None, eps=None, row_offset=None, _parts=None, rows=None):
        """torchmodel.py:302-327 -> (G, terms, ps1_mean, po1)"""
        return self.calculate_G(s0, pi0, 1, stage=stage, eps=eps, row_offset=row_offset, _mean_mode=True, _parts=_parts, rows=rows)

    def _rollout(self, o, pi, steps, calc_mean, samples, per_stage_mean, stage, eps, row_offset):
        e = self._ready()
        o = e.tensor(o, (-1, self.colour_channels, self.resolution, self.resolution)); pi = e.tensor(pi, (-1, self.pi_dim))
        M = o.shape[0]
        if pi.shape[0] != M:
            raise ValueError('o and pi must have the same number of rows')
        steps, samples = int(steps), int(samples)
        if steps < 1 or samples < 1:
            raise RuntimeError('efe engine: steps and samples must be >= 1')
        nz = self._noise(self._take_stage(stage, steps), 0, 0, row_offset)
        S_eff = 1 if (per_stage_mean and calc_mean) else samples
        if eps is None and self.eps_source is not None:
            parts = [self._src_eps(M, 10, PASS_ROOT, 0, nz.stage, row_offset).reshape(-1)]
            parts += [self._src_eps_calcG(M, S_eff, nz.stage + t, row_offset).reshape(-1) for t in range(steps)]
            eps = np.concatenate(parts)
        eps_t = e.tensor(eps).reshape(-1) if eps is not None else None
        if eps_t is not None and eps_t.numel() != M * 10 + steps * 3 * S_eff * M * 10:
            raise ValueError(f'eps has {eps_t.numel()} elements, efe_rollout expects M*10 + steps*3*S*M*10 = '
                             f'{M * 10 + steps * 3 * S_eff * M * 10}')
        sum_G, sum_terms, po1 = e.ops.rollout(e.h, o, pi, steps, samples, bool(calc_mean), bool(per_stage_mean), self._seed64(), nz.stage,
                                              nz.row_offset, eps_t)
        return sum_G, [sum_terms[0], sum_terms[1], sum_terms[2]], po1

    def calculate_G_repeated(self, o, pi, steps=1, calc_mean=False, samples=10, *, stage=None, eps=None, row_offset=None):
        """torchmodel.py:227-245 -> (sum_G, sum_terms, po1); one row = one EFE rollout."""
        return self._rollout(o, pi, steps, calc_mean, samples, False, stage, eps, row_offset)

    def calculate_G_4_repeated(self, o, steps=1, calc_mean=False, samples=10, *, stage=None, eps=None, row_offset=None):
        """torchmodel.py:247-268 (4 rows, pi = eye(4); calc_mean switches every stage to calculate_G_mean)"""
        if self.pi_dim != 4:
            raise ValueError('calculate_G_4_repeated is hard-wired to 4 actions (pi_one_hot, torchmodel.py:251-252)')
        o = self._engine.tensor(o, (-1, self.colour_channels, self.resolution, self.resolution))
        if o.shape[0] != 4:
            raise ValueError('calculate_G_4_repeated expects 4 rows (torchmodel.py:251-252)')
        return self._rollout(o, self.pi_one_hot, steps, calc_mean, samples, True, stage, eps, row_offset)

    def calculate_G_given_trajectory(self, s0_traj, ps1_traj, ps1_mean_traj, ps1_logvar_traj, pi0_traj, *, stage=None,
                                     eps=None, row_offset=None):
        """torchmodel.py:329-352 -> G[T]"""
        e = self._ready()
        s0 = e.tensor(s0_traj, (-1, 10)); ps1 = e.tensor(ps1_traj, (-1, 10)); mean = e.tensor(ps1_mean_traj, (-1, 10))
        lv = e.tensor(ps1_logvar_traj, (-1, 10)); pi0 = e.tensor(pi0_traj, (-1, self.pi_dim))
        T = s0.shape[0]
        nz = self._noise(stage, 0, 0, row_offset)
        if eps is None and self.eps_source is not None:
            eps = self._src_eps_calcG(T, 1, nz.stage, row_offset)      # the T1 slot is unused in trajectory mode
        eps_t = e.tensor(eps, (3, T, 10)) if eps is not None else None
        return e.ops.trajectory(e.h, s0, ps1, mean, lv, pi0, self._seed64(), nz.stage, nz.row_offset, eps_t)

    def simulate_batch(self, starting_s, depth, use_means=False, *, stage=None, row_offset=None, eps=None, u=None, rows=None):
        """mcts_step_simulate for E lock-step episodes -> (G[E], pi0[E,depth,4], Qpi0[E,4]).
        eps: optional injected normals, flat [depth*E*10 (step transitions)] + [3*E*depth*10 (trajectory T1/T2/D2B)];
        u: optional injected action uniforms [depth, E]; rows: optional Rows (entry = episode)."""
        e = self._ready()
        s = e.tensor(starting_s, (-1, 10))
        E, T = s.shape[0], int(depth)
        nz = self._noise(stage, 0, 0, row_offset)
        ro = self.row_offset if row_offset is None else int(row_offset)
        src_stage = nz.stage
        injected = (eps is None and self.eps_source is not None) or (u is None and self.u_source is not None)
        he = rows.host_rows(1) if (rows is not None and injected) else None          # a compacted call with injected noise: the episodes it holds
        if eps is None and self.eps_source is not None:
            if he is None:
                parts = [self._src_eps(E, 10, PASS_SIM, t, src_stage, ro).reshape(-1) for t in range(T)]
                parts.append(self._src_eps_calcG(E * T, 1, src_stage, ro * T).reshape(-1))
            else:
                n, ht = int(he.max()) + 1, (he[:, None] * T + np.arange(T)[None]).reshape(-1)
                parts = [self._src_eps(n, 10, PASS_SIM, t, src_stage, ro)[he].reshape(-1) for t in range(T)]
                parts.append(self._src_eps_calcG(n * T, 1, src_stage, ro * T)[:, ht].reshape(-1))
            eps = np.concatenate(parts)
        if u is None and self.u_source is not None:
            n = E if he is None else int(he.max()) + 1
            u = np.stack([np.asarray(self.u_source(self.seed, n, PASS_HABIT, t, src_stage, ro), dtype=np.float32) for t in range(T)], 0)
            if he is not None:
                u = u[:, he]
        eps_t = e.tensor(eps).reshape(-1) if eps is not None else None
        if eps_t is not None and eps_t.numel() != 4 * T * E * 10:
            raise ValueError(f'eps has {eps_t.numel()} elements, efe_simulate expects depth*E*10 + 3*E*depth*10 = {4 * T * E * 10}')
        u_t = e.tensor(u, (T, E)) if u is not None else None
        rm, ri, rnt = (rows.mask, rows.ids, rows.n_total) if rows is not None else (None, None, 0)
        return e.ops.simulate(e.h, s, T, bool(use_means), self._seed64(), nz.stage, nz.row_offset, eps_t, u_t, rm, ri, rnt)

    def mcts_step_simulate(self, starting_s, depth, use_means=False, *, stage=None, row_offset=None, eps=None, u=None):
        """torchmodel.py:354-393 -> (float G, pi0[depth,4], Qpi[4])"""
        G, pi0, q0 = self.simulate_batch(self._engine.tensor(starting_s, (1, 10)), depth, use_means, stage=stage, row_offset=row_offset,
                                         eps=eps, u=u)
        return G[0].item(), pi0[0], q0[0]

    def action_posterior(self, sum_G, single_values=4, temperature=10.0):
        """softmax_multi_with_log(-sum_G, 4) (/root/reference/src/util.py:46-53,68) -> (P, logP) [n,4] on device"""
        e = self._ready()
        return e.ops.action_posterior(e.h, e.tensor(sum_G).reshape(-1), int(single_values), float(temperature))

    # (generic geometry: convT1_generic + dec_a_... = ConvT1, ConvT2 as one launch each -- or, by default, only dec_a_... = both layers in k_convt_12;
    #  dec_b_... = ConvT3 (+ the final layer when fused); final_layer_generic)
    PROF_CLASSES = ('transition_mlp', 'dec_dense_small', 'dec_dense_16384', 'convT1_generic', 'dec_a_convT1_convT2',
                    'dec_b_convT3_final_reduce', 'final_layer_generic', 'encoder', 'other')

    def generic_class_names(self):
        """PROF_CLASSES name -> kernel description for the launches of the generic-geometry path (bench.py per-class table)"""
        names = {'dec_dense_16384': 'k_fc4 (Linear 256 -> 64 base^2)', 'convT1_generic': 'k_convt_p<1> (ConvT 64->64 s1)',
                 'dec_a_convT1_convT2': 'k_convt_p<2> (ConvT 64->64 s2)', 'dec_b_convT3_final_reduce': 'k_convt_p<2> (ConvT 64->32 s2)',
                 'final_layer_generic': 'k_final_g (ConvT 32->C + sigmoid + reductions)', 'encoder': 'encoder (k_conv_e12 layers 1-2, k_conv_g layers 3-4, dense head)',
                 'transition_mlp': 'k_trans_fused', 'dec_dense_small': 'k_head (decoder head 10 -> 256 -> 256 -> 256, one launch)'}
        if getattr(self, '_opts', {}).get('fuse_final_g', 1) and self.resolution != 32:
            names['dec_b_convT3_final_reduce'] = 'k_dec_bg (ConvT 64->32 s2 + ReLU + ConvT 32->C + sigmoid + per-image sums, fused)'
            del names['final_layer_generic']
        return names

    def prof_enable(self, on=True, classes=None):
        """time kernel classes with HIP events on the launch stream; `classes` = iterable of PROF_CLASSES names
        (default: all)"""
        e = self._ready()
        mask = 0
        if on:
            mask = -1 if classes is None else sum(1 << self.PROF_CLASSES.index(c) for c in classes)
        e.check(e.lib.efe_prof_enable(e.ctx, mask))
        self._prof_on = bool(mask)

    def prof_read(self):
        """-> {class name: (milliseconds, launches)} since the last read (synchronises)."""
        e = self._ready()
        n = e.lib.efe_prof_classes()
        ms = (C.c_double * n)(); cnt = (C.c_int64 * n)()
        e.check(e.lib.efe_prof_read(e.ctx, ms, cnt))
        return {self.PROF_CLASSES[i]: (ms[i], cnt[i]) for i in range(n)}

    def last_call_macs(self):
        return int(self._engine.lib.efe_last_call_macs(self._engine.ctx))
