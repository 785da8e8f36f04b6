"""Where the variational state of an `HPF` object lives between calls.

The reference keeps its eight state arrays (Theta, Beta, Gamma_shp, Gamma_rte, Lambda_shp, Lambda_rte, k_rte,
t_rte) as numpy attributes and every method mutates them in place (hpfrec/__init__.py: partial_fit INIT:914-927,
add_user INIT:1145-1196, topN INIT:1337 ...).  On the device path a `partial_fit` call touches every row of the batch
side's tables (PXI:443-473), so shipping the state up and down per call costs gigabytes over PCIe for a batch of a few
thousand triplets.  Here each array has a host copy and a device copy with explicit validity flags:

  * in-package mutations (partial_fit, fold-in) run on the device copy and mark the host copy stale;
  * reading the public attribute (`model.Beta`) brings the host copy up to date -- in place, so older references
    stay valid -- and HANDS THE ARRAY OUT: from then on the caller holds a reference and may edit it in place at any
    time, as the reference allows, so while the host copy is the current one every device operation re-uploads that
    table first (a 100 MB table costs a few ms; correctness over speed).  Assigning the attribute hands out too (the
    caller keeps the array it passed).  The device copy is trusted again once the package itself has rewritten the
    table on the device (the host copy is then stale until somebody asks for it);
  * package code reads through `peek_host` / `rows` / `table`, which do not hand anything out: an object that is
    only fitted / partial_fit / queried keeps its state on the device and moves batches and results only.

Nothing is inferred from array contents (no fingerprints): a table is re-used on the device only while no reference
to its host copy can exist outside the package.  `version` counts every change of either copy.
"""
import numpy as np
import torch

from . import _lib

NAMES = ("Theta", "Beta", "Gamma_shp", "Gamma_rte", "Lambda_shp", "Lambda_rte", "k_rte", "t_rte")
_USER_SIDE = ("Theta", "Gamma_shp", "Gamma_rte", "k_rte")


class ResidentState:
    def __init__(self):
        self.host = {}         # name -> ndarray (or None)
        self.host_ok = {}      # host copy is current
        self.dev_ok = {}       # device copy (self.model's table) is current
        self.handed = {}       # a reference to the host array exists outside the package (see module docstring)
        self.model = None      # svi.DeviceModel
        self.version = 0
        self.stats = {"h2d_bytes": 0, "d2h_bytes": 0}

    # -- host side ------------------------------------------------------------------------------
    def has(self, name):
        return name in self.host

    def set_host(self, name, arr, private=False):
        """private=True: the package created `arr` and keeps the only reference."""
        self.host[name] = arr
        self.host_ok[name] = True
        self.dev_ok[name] = False
        self.handed[name] = not private
        self.version += 1

    def drop(self, name):
        for d in (self.host, self.host_ok, self.dev_ok, self.handed):
            d.pop(name, None)
        self.version += 1

    def peek_host(self, name):
        """Current host copy, for package code that only reads it."""
        if not self.host_ok.get(name, False):
            out = self.host.get(name)
            got = self.model.get(name)
            self.stats["d2h_bytes"] += got.nbytes
            if isinstance(out, np.ndarray) and out.shape == got.shape and out.dtype == got.dtype and out.flags.writeable:
                out[...] = got           # in place: references handed out earlier stay current
            else:
                self.host[name] = got
                self.handed[name] = False     # a fresh array: nobody else has it
            self.host_ok[name] = True
        return self.host[name]

    def get_host(self, name):
        """The public attribute: the caller now holds the array and may edit it in place whenever it likes."""
        arr = self.peek_host(name)
        if arr is not None:
            self.handed[name] = True
            self.version += 1
        return arr

    def _device_current(self, name):
        """The device copy may be used as it is: it is up to date and no outside reference to a CURRENT host copy
        exists (a stale host copy cannot be the source of truth, whoever holds it)."""
        return self.dev_ok.get(name, False) and not (self.host_ok.get(name, False) and self.handed.get(name, False))

    # -- device side ----------------------------------------------------------------------------
    def _shapes(self):
        th, be = self.host.get("Theta"), self.host.get("Beta")
        if th is None or be is None:
            if self.model is not None:
                return self.model.nU, self.model.nI, self.model.k
            raise ValueError("the model has no Theta/Beta yet")
        return int(th.shape[0]), int(be.shape[0]), int(th.shape[1])

    def ensure_model(self, ops, names=NAMES, lazy_ok=False):
        """The DeviceModel with the listed tables current (uploads the stale ones).  lazy_ok: the caller is a stochastic
        step, which works on the factored / stale forms its predecessor left (svi.DeviceModel.materialize); everybody
        else gets the tables themselves."""
        from . import svi
        nU, nI, k = self._shapes()
        m = self.model
        if m is None or (m.nU, m.nI, m.k) != (nU, nI, k) or m.ops.device != ops.device or m.ops is not ops:
            for n in self.host:            # shape change (users/items appended) or first use: everything goes up
                if not self.host_ok.get(n, False) and m is not None:
                    self.peek_host(n)
                self.dev_ok[n] = False
            self.model = m = svi.DeviceModel(ops, k, nU, nI)
        for n in names:
            if n in self.host and self.host[n] is not None and not self._device_current(n):
                a = self.host[n]
                want = (nU if n in _USER_SIDE else nI)
                if a.shape[0] != want:
                    raise ValueError("%s has %d rows, expected %d" % (n, a.shape[0], want))
                m.put(n, a)
                self.stats["h2d_bytes"] += int(a.nbytes)
                self.dev_ok[n] = True
        if not lazy_ok:
            m.materialize()
        return m

    def adopt(self, model, names=NAMES):
        """`model` (a DeviceModel of this state's shape, built on the ops object later calls will use) holds the
        CURRENT values of `names` -- a fit that finished on the device.  The host copies become stale; they are
        brought up to date, in place, when somebody reads them."""
        self.model = model
        for n in names:
            self.host.setdefault(n, None)
            self.handed.setdefault(n, False)
            self.host_ok[n] = False
            self.dev_ok[n] = True
        self.version += 1

    def touched(self, names=NAMES):
        """The device copies of `names` were just mutated."""
        for n in names:
            if n in self.host:
                self.host_ok[n] = False
                self.dev_ok[n] = True
        self.version += 1

    def table(self, ops, name):
        """Padded device table [rows][ld] of one array, current."""
        return getattr(self.ensure_model(ops, (name,)), name)

    def on_device(self, name):
        return self.model is not None and self._device_current(name)

    def device_row(self, name, idx):
        """Padded device row [ld] of a table whose device copy is current, else None (never uploads)."""
        if not self.on_device(name) or idx < 0:
            return None
        self.model.materialize()
        return getattr(self.model, name)[int(idx)]

    def rows(self, name, idx):
        """Host copy of a few rows, from whichever copy is current, without handing the array out."""
        idx = np.atleast_1d(np.asarray(idx, dtype=np.int64))
        if self.host_ok.get(name, False):
            return self.host[name][idx]
        m = self.model
        m.materialize()
        n = getattr(m, name).shape[0]
        t = torch.from_numpy(np.where(idx < 0, idx + n, idx)).to(m.ops.device)
        if name in ("k_rte", "t_rte"):
            return getattr(m, name)[t].cpu().numpy().reshape(-1, 1)
        return getattr(m, name)[t][:, : m.k].contiguous().cpu().numpy()

    # -- pickling: host arrays only ---------------------------------------------------------------
    def __getstate__(self):
        for n in list(self.host):
            if not self.host_ok.get(n, False):
                self.peek_host(n)
        return {"host": dict(self.host), "version": self.version}

    def __setstate__(self, st):
        self.__init__()
        for n, a in st["host"].items():
            self.set_host(n, a, private=True)
        self.version = st.get("version", 0)


class StateArray:
    """Descriptor: `HPF.Theta` etc. are views of the object's ResidentState (attribute semantics of the reference:
    plain arrays one can read, edit in place or re-assign; AttributeError until first assigned)."""

    def __init__(self, name):
        self.name = name

    def __get__(self, obj, objtype=None):
        if obj is None:
            return self
        st = obj.__dict__.get("_state")
        if st is None or not st.has(self.name):
            raise AttributeError(self.name)
        return st.get_host(self.name)

    def __set__(self, obj, value):
        st = obj.__dict__.get("_state")
        if st is None:
            st = obj.__dict__["_state"] = ResidentState()
        st.set_host(self.name, value)

    def __delete__(self, obj):
        st = obj.__dict__.get("_state")
        if st is None or not st.has(self.name):
            raise AttributeError(self.name)
        st.drop(self.name)


def ld_for(k):
    return _lib.ld_for_k(int(k))


class DeviceBackedArray:
    """Descriptor for a plain host-array attribute of the reference (`HPF.seen`) that the package produces on the
    device: it stays there until somebody reads the attribute (then it is downloaded once and cached); assigning the
    attribute -- the class itself does, INIT:1186-1196 -- drops the device copy.  Package code that can work on the
    device copy asks `device_of(obj)`."""

    def __init__(self, name):
        self.name, self.slot = name, "_devbacked_" + name

    def __get__(self, obj, objtype=None):
        if obj is None:
            return self
        cell = obj.__dict__.get(self.slot)
        if cell is None:
            raise AttributeError(self.name)
        if cell[0] is None:
            cell[0] = cell[1].cpu().numpy()
        # handed out: the caller may edit the host array in place at any time, so the device copy is no longer trusted
        # (the same rule ResidentState applies to the state tables); device_of() answers None from now on
        cell[1] = None
        return cell[0]

    def __set__(self, obj, value):
        obj.__dict__[self.slot] = [value, None]

    def __delete__(self, obj):
        if obj.__dict__.pop(self.slot, None) is None:
            raise AttributeError(self.name)

    def set_device(self, obj, tensor):
        """The attribute's value, as a device tensor of the dtype the host array must have."""
        obj.__dict__[self.slot] = [None, tensor]

    def device_of(self, obj):
        cell = obj.__dict__.get(self.slot)
        return None if cell is None else cell[1]
