"""Packed valid-cell layout of the L x L proposal map.

The reference keeps dense (B, L, L, ...) tensors whose masked cells are exactly zero after every
layer; the kernels instead work on the sorted list of cells that are present.  Two lists are used:
``from_mask`` (cells with moment_mask == 1; the in-model fast path) and ``all_cells`` (every
(b, i, j) with its mask flag; reproduces the dense sub-module seams for arbitrary inputs)."""
import torch


class CellLayout:
    __slots__ = ("cells", "row_ptr", "cellmap", "N", "B", "L", "_idx", "all_valid")

    def __init__(self, cells, row_ptr, cellmap, B, L, idx, all_valid=False):
        self.cells, self.row_ptr, self.cellmap = cells, row_ptr, cellmap
        self.all_valid = all_valid                                # mask-driven list: every listed cell has m == 1
        self.N, self.B, self.L = int(cells.shape[0]), B, L
        self._idx = idx                                           # (N, 3) int64, or None: derived from cells on first use

    def _index(self):
        if self._idx is None:
            self._idx = self.cells[:, :3].long()
        return self._idx

    bidx = property(lambda self: self._index()[:, 0])
    iidx = property(lambda self: self._index()[:, 1])
    jidx = property(lambda self: self._index()[:, 2])

    @staticmethod
    def _build(present, flag, N=None):
        B, L, _ = present.shape
        if present.is_cuda and (N is not None or present is not flag):
            # HIP path: two launches (csrc/layout.hip).  all_cells lists every (b, i, j): N = B*L*L without asking the device
            from . import _lib
            all_cells = present is not flag
            N = B * L * L if all_cells else N
            mask = flag.contiguous()
            mask = mask.view(torch.uint8) if mask.dtype == torch.bool else (mask != 0).view(torch.uint8)
            cells = torch.empty((N, 4), dtype=torch.int32, device=flag.device)
            row_ptr = torch.empty(B * L + 1, dtype=torch.int32, device=flag.device)
            cellmap = torch.empty((B, L, L), dtype=torch.int32, device=flag.device)
            _lib.call("smin_build_cells", _lib.stream(), _lib.ptr(mask), B, L, int(all_cells), _lib.ptr(cells), _lib.ptr(row_ptr), _lib.ptr(cellmap))
            return CellLayout(cells, row_ptr, cellmap, B, L, None, not all_cells)
        if N is None:
            idx = present.nonzero()                               # (N, 3) sorted by (b, i, j); one host sync
        else:
            idx = torch.nonzero_static(present, size=N)           # count already known: no sync
        N = idx.shape[0]
        m = flag[idx[:, 0], idx[:, 1], idx[:, 2]].to(torch.int32).unsqueeze(1)
        cells = torch.cat([idx.to(torch.int32), m], dim=1).contiguous()
        counts = present.sum(dim=2).reshape(-1)
        row_ptr = torch.zeros(B * L + 1, dtype=torch.int32, device=present.device)
        row_ptr[1:] = torch.cumsum(counts, 0).to(torch.int32)
        cellmap = torch.full((B, L, L), -1, dtype=torch.int32, device=present.device)
        cellmap[idx[:, 0], idx[:, 1], idx[:, 2]] = torch.arange(N, dtype=torch.int32, device=present.device)
        return CellLayout(cells, row_ptr, cellmap, B, L, idx, present is flag)

    @staticmethod
    def from_mask(moment_mask):
        mm = moment_mask != 0
        return CellLayout._build(mm, mm)

    @staticmethod
    def all_cells(moment_mask):
        mm = moment_mask != 0
        return CellLayout._build(torch.ones_like(mm), mm)

    @staticmethod
    def begin(moment_mask):
        """Start building the mask-driven layout without blocking the host: the cell count is copied to pinned
        memory asynchronously; call .finish() after enqueueing independent work (the backbone)."""
        return _PendingLayout(moment_mask)

    # dense (B, L, L, ...) <-> packed (N, ...) through torch indexing (differentiable; seams only)
    def pack(self, dense):
        return dense[self.bidx, self.iidx, self.jidx].contiguous()

    def unpack(self, packed):
        out = packed.new_zeros((self.B, self.L, self.L) + tuple(packed.shape[1:]))
        out[self.bidx, self.iidx, self.jidx] = packed
        return out


class _PendingLayout:
    def __init__(self, moment_mask):
        self.mm = moment_mask != 0
        self.event = None
        if self.mm.is_cuda:
            self.host = torch.empty(1, dtype=torch.int64, pin_memory=True)
            self.host.copy_(self.mm.sum().reshape(1), non_blocking=True)
            self.event = torch.cuda.Event()
            self.event.record()

    def finish(self):
        if self.event is None:
            return CellLayout._build(self.mm, self.mm)
        self.event.synchronize()                                  # waits only for work queued before the count
        return CellLayout._build(self.mm, self.mm, int(self.host[0]))
