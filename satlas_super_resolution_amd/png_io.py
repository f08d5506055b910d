"""PNG decode / encode workers of the inference drivers (numpy + Pillow only: the module is the program of the worker
processes, which must not pay for - or touch - torch and the HIP runtime).

Why processes: Pillow holds the GIL for most of a small PNG's decode / encode, so the thread pool the drivers used first did
not scale at all (3 tiles of 256 chunks on 8 cores: 1.44 tiles/s with one thread, 1.01 with eight; the device was busy 2-13 %
of the time at 2.9 tiles/s on the GPU box, profiles/r03j_infer_e2e_*.json).  Same files, same pixels - the reference's
infer_grid.py:55-85 reads and writes them one by one with skimage.io."""
from __future__ import annotations

import os
import pickle
import queue
import subprocess
import sys
import threading
from concurrent.futures import Future
from typing import List, Sequence, Tuple

import numpy as np


def host_cores() -> int:
    """cores this process may actually run on: the affinity mask, capped by the cgroup CPU quota (os.cpu_count() reports the
    machine's CPUs even inside a container limited to a few)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(float(q) / float(per))))
    except (OSError, ValueError):
        pass
    return n


def read_png(path: str) -> np.ndarray:
    from PIL import Image
    with Image.open(path) as im:
        return np.array(im.convert("RGB"))


def read_many(paths: Sequence[str]) -> List[np.ndarray]:
    return [read_png(p) for p in paths]


def encode_png(arr: np.ndarray, level: int = 1) -> bytes:
    """A [H, W, 3] uint8 image as an 8-bit RGB PNG: filter type 0 ("None") on every scanline, one IDAT chunk, zlib level `level`.
    Same pixels as any other encoder's file; 4-6x cheaper than Pillow's writer, which tries the five row filters on every
    scanline before zlib's default level 6 (2.7 ms per 128 x 128 chunk against 0.45 ms here; the 2048 x 2048 mosaic 160 ms
    against 70 ms) - the PNG encode was what kept whole-tile inference host-bound (profiles/r03io_*.json)."""
    import struct
    import zlib
    a = np.ascontiguousarray(arr, dtype=np.uint8)
    assert a.ndim == 3 and a.shape[2] == 3, a.shape
    h, w = a.shape[:2]
    rows = np.empty((h, 1 + 3 * w), np.uint8)
    rows[:, 0] = 0                                    # filter type of the scanline: None
    rows[:, 1:] = a.reshape(h, 3 * w)

    def chunk(tag: bytes, data: bytes) -> bytes:
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)
    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) +
            chunk(b"IDAT", zlib.compress(rows.tobytes(), level)) + chunk(b"IEND", b""))


def save_png(arr: np.ndarray, path: str) -> None:
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "wb") as f:
        f.write(encode_png(arr))


def save_many(items: Sequence[Tuple[np.ndarray, str]]) -> int:
    for arr, path in items:
        save_png(arr, path)
    return len(items)


def mosaic(cells: Sequence[np.ndarray], img_size: int, grid_size: int = 16, sentinel2: bool = False) -> np.ndarray:
    """infer_utils.stitch (/root/reference/ssr/utils/infer_utils.py:41-60) on in-memory chunks given in row-major cell order:
    cell (i, j) lands at rows i*cs.., columns j*cs.. with cs = int(img_size / grid_size); [n*32, 32, 3] Sentinel-2 stacks
    contribute their first frame when sentinel2=True; a margin stays zero when img_size is not divisible by grid_size."""
    cs = int(img_size / grid_size)
    first = (lambda a: np.reshape(a, (-1, 32, 32, 3))[0]) if sentinel2 else (lambda a: a)
    tiles = np.stack([first(np.asarray(c)) for c in cells])
    assert tiles.shape == (grid_size * grid_size, cs, cs, 3), (tiles.shape, cs)
    m = tiles.reshape(grid_size, grid_size, cs, cs, 3).transpose(0, 2, 1, 3, 4).reshape(grid_size * cs, grid_size * cs, 3)
    canvas = np.zeros((img_size, img_size, 3), np.uint8)
    canvas[:grid_size * cs, :grid_size * cs] = m
    return canvas


def stitch_and_save(cells: Sequence[np.ndarray], img_size: int, path: str, sentinel2: bool = False) -> str:
    save_png(mosaic(cells, img_size, 16, sentinel2), path)
    return path


def stitch_from_dir(chunks_dir: str, img_size: int, path: str, sentinel2: bool = False) -> str:
    """the reference's stitch(): re-reads `{chunks_dir}/{i}_{j}.png` (several ranks wrote them)"""
    cells = [read_png(os.path.join(chunks_dir, f"{i}_{j}.png")) for i in range(16) for j in range(16)]
    return stitch_and_save(cells, img_size, path, sentinel2)


# ---- shared-memory transport (round 4).  Pixels do not travel through the pipes: the driver and the workers map the same files
# (np.memmap of files under /dev/shm, or under the temporary directory when /dev/shm is small), a task names a block, offsets and
# shapes.  With pickled arrays the driver thread moved ~37 MB per tile through 64-KB pipe buffers under the GIL and whole-tile
# inference stayed at 4.4 tiles/s whatever the codecs cost (profiles/r04j_*.json).
class ShmBlock:
    """a file-backed shared byte block; the creating side owns (and finally unlinks) the file"""

    _serial = [0]

    def __init__(self, nbytes: int, directory: str, tag: str):
        self.nbytes = int(nbytes)
        ShmBlock._serial[0] += 1           # never the same path twice: the (long-lived) workers cache their mappings by path
        self.path = os.path.join(directory, f"ssr_png_{os.getpid()}_{ShmBlock._serial[0]}_{tag}")
        import mmap
        with open(self.path, "wb") as f:
            f.truncate(self.nbytes)
        self._f = open(self.path, "r+b")
        self._mm = mmap.mmap(self._f.fileno(), self.nbytes)
        self.buf = np.frombuffer(self._mm, dtype=np.uint8)       # a plain ndarray: slicing a np.memmap costs ~10 us per view

    def close(self):
        self.buf = None
        try:
            self._mm.close()                                   # refused while a view of the block is still alive somewhere:
        except (BufferError, ValueError):                      # then the mapping goes with the last view
            pass
        try:
            self._f.close()
            os.unlink(self.path)
        except OSError:
            pass


def shm_dir(need_bytes: int) -> str:
    import shutil
    import tempfile
    try:
        if shutil.disk_usage("/dev/shm").free >= 2 * need_bytes:
            return "/dev/shm"
    except OSError:
        pass
    return tempfile.gettempdir()


_MAPS = {}          # path -> (ndarray view, mmap): the FIXED blocks of a run (in / out rings), mapped once per worker
_MAPS_MAX = 16


def _open_map(path: str, nbytes: int):
    import mmap
    with open(path, "r+b") as f:
        mm = mmap.mmap(f.fileno(), nbytes)
    return [np.frombuffer(mm, dtype=np.uint8), mm]             # a LIST: _close_map takes the view out of it before closing


def _close_map(entry) -> bool:
    """unmap a block: the entry's own view is dropped first (an ndarray export keeps mmap.close() from succeeding - with the view
    still referenced from the entry the close used to fail every time and the pages of an unlinked block stayed allocated until the
    entry itself died).  Returns whether the mapping is closed; False = a caller still holds a view, the mapping goes with it."""
    mm = entry[1]
    entry[0] = None
    try:
        mm.close()                                             # gives the pages of an unlinked block back to the tmpfs
    except (BufferError, ValueError):
        return False
    return True


def _map(path: str, nbytes: int) -> np.ndarray:
    """cached mapping of a long-lived block.  Least-recently-used entries are CLOSED when the cache is full: a worker that merely
    dropped its reference kept the tmpfs pages of blocks the driver had long unlinked (a 12.6-MB mosaic block per tile and worker
    on a multi-hundred-tile run filled /dev/shm; one-shot blocks now go through `transient`)."""
    e = _MAPS.pop(path, None)
    if e is not None and e[0].shape[0] != nbytes:
        _close_map(e)
        e = None
    if e is None:
        while len(_MAPS) >= _MAPS_MAX:
            _close_map(_MAPS.pop(next(iter(_MAPS))))
        e = _open_map(path, nbytes)
    _MAPS[path] = e                                            # (re-)inserted last = most recently used
    return e[0]


def read_into(paths: Sequence[str], shm_path: str, nbytes: int, offsets: Sequence[int], slot_bytes: int, transient: bool = False):
    """decode every file into its slot of the block; returns the shapes (an image that does not fit its slot comes back as an array).
    transient: the block is used once (map, fill, unmap; nothing cached)"""
    e = _open_map(shm_path, nbytes) if transient else None
    m = e[0] if transient else _map(shm_path, nbytes)
    out = []
    for pth, off in zip(paths, offsets):
        a = read_png(pth)
        if a.nbytes > slot_bytes:
            out.append(a)
            continue
        m[off:off + a.nbytes] = a.reshape(-1)
        out.append(tuple(a.shape))
    if transient:
        del m
        _close_map(e)
    return out


def save_from(shm_path: str, nbytes: int, items: Sequence[Tuple[int, Tuple[int, ...], str]], transient: bool = False) -> int:
    """encode and write the images that lie at (offset, shape) in the block.  transient: a one-shot block (a tile's mosaic) -
    mapped, encoded and unmapped here, so that the driver's unlink really frees it"""
    e = _open_map(shm_path, nbytes) if transient else None
    m = e[0] if transient else _map(shm_path, nbytes)
    for off, shape, path in items:
        n = int(np.prod(shape))
        save_png(m[off:off + n].reshape(shape), path)
    if transient:
        del m
        _close_map(e)
    return len(items)


def timed(name: str, *args):
    """a task with its start / end times on the worker's clock (SSR_INFER_TRACE)"""
    import time
    t0 = time.time()
    res = _TASKS[name](*args)
    return (t0, time.time(), os.getpid(), res)


_TASKS = {"read_many": read_many, "save_many": save_many, "stitch_and_save": stitch_and_save, "stitch_from_dir": stitch_from_dir,
          "read_into": read_into, "save_from": save_from, "timed": timed}


class PngWorkerPool:
    """`n` worker processes (`python -m satlas_super_resolution_amd.png_io`), each behind one feeder thread; tasks and results
    travel as pickles over the workers' stdin / stdout.  Plain subprocesses, not multiprocessing: no fork of a process that
    holds a HIP context, and no re-import of the caller's __main__ in the children (spawn / forkserver do that).
    n = 0: the tasks run on `threads` threads of this process instead."""

    def __init__(self, n: int, threads: int = 8):
        self.q: "queue.Queue" = queue.Queue()
        self.procs, self.threads = [], []
        env = dict(os.environ)
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        env["PYTHONPATH"] = root + (os.pathsep + env["PYTHONPATH"] if env.get("PYTHONPATH") else "")
        for _ in range(n):
            p = subprocess.Popen([sys.executable, "-m", "satlas_super_resolution_amd.png_io"], stdin=subprocess.PIPE,
                                 stdout=subprocess.PIPE, env=env)
            self.procs.append(p)
        for k in range(n if n > 0 else max(1, threads)):
            t = threading.Thread(target=self._serve, args=(self.procs[k] if n > 0 else None,), daemon=True)
            t.start()
            self.threads.append(t)

    def submit(self, name: str, *args) -> Future:
        assert name in _TASKS, name
        f: Future = Future()
        self.q.put((f, name, args))
        return f

    def _serve(self, proc):
        while True:
            item = self.q.get()
            if item is None:
                return
            f, name, args = item
            try:
                if proc is None:
                    f.set_result(_TASKS[name](*args))
                    continue
                pickle.dump((name, args), proc.stdin, protocol=pickle.HIGHEST_PROTOCOL)
                proc.stdin.flush()
                ok, res = pickle.load(proc.stdout)
                if ok:
                    f.set_result(res)
                else:
                    f.set_exception(RuntimeError(f"png_io worker: {name} failed\n{res}"))
            except BaseException as e:      # a dead worker must not leave the caller waiting for ever
                f.set_exception(e)

    def close(self):
        for _ in self.threads:
            self.q.put(None)
        for t in self.threads:
            t.join()
        for p in self.procs:               # every worker sees EOF first, then they exit side by side (one by one: 13 ms each)
            try:
                p.stdin.close()
            except OSError:
                pass
        for p in self.procs:
            p.wait()
        self.procs, self.threads = [], []

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


_SHARED = {}


class shared_pool:
    """`with shared_pool(n) as pool`: a PngWorkerPool that outlives the block - the next call with the same n gets the same
    workers (interpreter start + numpy / Pillow import of 15 processes is ~0.4 s, more than a whole tile takes); closed at exit.
    Workers end by themselves when this process goes away (EOF on their stdin)."""

    def __init__(self, n: int, threads: int = 8):
        self.key = (n, threads)

    def __enter__(self) -> "PngWorkerPool":
        pool = _SHARED.get(self.key)
        if pool is None or (pool.procs and any(p.poll() is not None for p in pool.procs)):
            if not _SHARED:
                import atexit
                atexit.register(close_shared_pools)
            pool = _SHARED[self.key] = PngWorkerPool(*self.key)
        return pool

    def __exit__(self, *exc):
        return False


def close_shared_pools():
    for pool in list(_SHARED.values()):
        pool.close()
    _SHARED.clear()


def _worker_main():
    inp, out = sys.stdin.buffer, sys.stdout.buffer
    sys.stdout = sys.stderr              # nothing but pickles on the pipe
    while True:
        try:
            name, args = pickle.load(inp)
        except EOFError:
            return
        try:
            res = (True, _TASKS[name](*args))
        except Exception:
            import traceback
            res = (False, traceback.format_exc())
        pickle.dump(res, out, protocol=pickle.HIGHEST_PROTOCOL)
        out.flush()


if __name__ == "__main__":
    _worker_main()
