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


def read_png(path: str) -> np.ndarray:
    from PIL import Image
    with Image.open(path) as im:
        return np.array(im.convert("RGB"))


def read_many(paths: Sequence[str]) -> List[np.ndarray]:
    return [read_png(p) for p in paths]


def save_png(arr: np.ndarray, path: str) -> None:
    from PIL import Image
    os.makedirs(os.path.dirname(path), exist_ok=True)
    Image.fromarray(np.ascontiguousarray(arr)).save(path)


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


_TASKS = {"read_many": read_many, "save_many": save_many, "stitch_and_save": stitch_and_save, "stitch_from_dir": stitch_from_dir}


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
        for p in self.procs:
            try:
                p.stdin.close()
            except OSError:
                pass
            p.wait()
        self.procs, self.threads = [], []

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


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
