"""Two ways to drive the C-ABI from the parity tests:

* ``EmuBackend``  -- the CPU wave-emulation build of the SAME kernel sources
  (tests/emu, test infrastructure), host numpy buffers.  Runs in the GPU-less
  container (-m "not gpu").
* ``GpuBackend``  -- the product library news_recommendation_amd/libnr_engine.so
  on cuda:0, torch device buffers (-m gpu).
"""
import ctypes
import os
import subprocess
import numpy as np

from news_recommendation_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class EmuBackend:
    name = 'emu'
    _lib = None

    def __init__(self):
        if EmuBackend._lib is None:
            so = os.path.join(ROOT, 'tests', 'emu', '_build', 'libnr_engine_emu.so')
            srcs = [os.path.join(ROOT, 'tests', 'emu', f) for f in ('nr_prims.h', 'nr_emu.cpp')]
            cs = os.path.join(ROOT, 'news_recommendation_amd', 'csrc')
            srcs += [os.path.join(cs, f) for f in os.listdir(cs) if f.endswith(('.h', '.hip'))]
            srcs.append(os.path.join(ROOT, 'include', 'nr_engine.h'))
            if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
                subprocess.check_call([os.path.join(ROOT, 'tests', 'emu', 'build_emu.sh')])
            EmuBackend._lib = _capi.bind(ctypes.CDLL(so))
        self.lib = EmuBackend._lib
        self.stream = None
        self._keep = []

    def dev(self, a):
        h = np.array(a, order='C', copy=True)
        self._keep.append(h)          # keep temporaries alive until the call that uses their pointer has run
        return h

    def empty(self, shape, dtype):
        return np.zeros(shape, dtype=dtype)

    def poison(self, shape, dtype):
        a = np.empty(shape, dtype=dtype)
        a.view(np.uint8)[...] = 0xFF
        return a

    def ptr(self, h):
        return None if h is None else h.ctypes.data

    def np(self, h):
        return np.array(h)

    def sync(self):
        self._keep.clear()


class GpuBackend:
    name = 'gpu'

    def __init__(self):
        import torch
        self.torch = torch
        assert torch.cuda.is_available()
        self.lib = _capi.load()
        self.device = torch.device('cuda:0')
        self._keep = []

    @property
    def stream(self):
        return self.torch.cuda.current_stream().cuda_stream

    _map = {np.dtype(np.float32): 'float32', np.dtype(np.int64): 'int64', np.dtype(np.int32): 'int32',
            np.dtype(np.uint16): 'int16'}

    def dev(self, a):
        a = np.ascontiguousarray(a)
        if a.dtype == np.uint16:
            a = a.view(np.int16)
        h = self.torch.from_numpy(a).to(self.device)
        self._keep.append(h)
        return h

    def empty(self, shape, dtype):
        return self.torch.zeros(shape, dtype=getattr(self.torch, self._map[np.dtype(dtype)]), device=self.device)

    def poison(self, shape, dtype):
        t = self.empty(shape, dtype)
        t.view(self.torch.uint8).fill_(0xFF)
        return t

    def ptr(self, h):
        return None if h is None else h.data_ptr()

    def np(self, h):
        a = h.detach().cpu().numpy()
        return a.view(np.uint16) if a.dtype == np.int16 else a

    def sync(self):
        self.torch.cuda.synchronize()
        self._keep.clear()


def bf16_to_f32(u):
    return (u.astype(np.uint32) << 16).view(np.float32)


def f32_to_bf16(x):
    """Round-to-nearest-even bf16 bits of a float32 array (matches v_cvt_pk_bf16_f32)."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    r = u + 0x7FFF + ((u >> 16) & 1)
    return (r >> 16).astype(np.uint16)


def bf16_round(x):
    return bf16_to_f32(f32_to_bf16(x))
