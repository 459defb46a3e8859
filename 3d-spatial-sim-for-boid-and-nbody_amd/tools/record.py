"""Offline recorder on the HIP backend: the reference's frame / state on-disk format
(tools/record.py) and its record() GPU loop (:702-935, GPU branch :821-832).

On-disk contract kept so the reference's playback can read the output:
  recordings/<session>/metadata.json                     (:50-58)
  frame_%04d.npz   = np.savez(positions=f32 (N,3), colors=f32 (N,3))          (:88-96)
  frame_%04d.zstd  = u8 format (1 absolute | 2 delta) + u32 len + zstd(positions) +
                     u32 len + zstd(colors); format 2 payload = int16((cur-prev)*1000)   (:231-326)
  state_%04d.npz   = positions, velocities every 50 frames, previous one deleted   (:867-876)
                     (+ `masses`, a superset key: the reference loses masses on resume [quirk])
The interactive menus, progress bars and the background compressor thread are UX and are not
reproduced; ``compress_recording`` converts finished .npz frames to .zstd with the same delta
chaining, and ``record(config with "zstd": True)`` writes .zstd frames directly: the int16 delta
payload is quantised ON THE DEVICE against the previous decoded frame kept in HBM (12 instead of
24 bytes per body over PCIe), zstd runs on the host.  ``extend_recording`` is the reference's
``--extend`` (:1156-1199); Ctrl-C leaves a state checkpoint like the reference (:916-935).  zstd comes from the system libzstd through ctypes (python-zstandard is
not installed in this image); without it the .zstd functions raise and raw .npz still works.
"""
import ctypes as C
import json
import os
import struct
import time
from datetime import datetime
from pathlib import Path

import numpy as np

PROJECT_ROOT = Path(__file__).resolve().parent.parent
COMPRESSION_BATCH_SIZE = 50  # reference :225
ZSTD_LEVEL = 19              # reference :252
STATE_EVERY = 50             # reference :867


# ---- zstd through ctypes ------------------------------------------------------------------
_zstd = None


def _load_zstd():
    global _zstd
    if _zstd is None:
        for name in ("libzstd.so.1", "libzstd.so"):
            try:
                z = C.CDLL(name)
                break
            except OSError:
                z = None
        if z is None:
            raise RuntimeError("libzstd not found: .zstd frames unavailable (raw .npz frames still work)")
        z.ZSTD_compressBound.restype = C.c_size_t
        z.ZSTD_compressBound.argtypes = [C.c_size_t]
        z.ZSTD_compress.restype = C.c_size_t
        z.ZSTD_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
        z.ZSTD_decompress.restype = C.c_size_t
        z.ZSTD_decompress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        z.ZSTD_getFrameContentSize.restype = C.c_ulonglong
        z.ZSTD_getFrameContentSize.argtypes = [C.c_void_p, C.c_size_t]
        z.ZSTD_isError.restype = C.c_uint
        z.ZSTD_isError.argtypes = [C.c_size_t]
        _zstd = z
    return _zstd


def zstd_compress(data: bytes, level: int = ZSTD_LEVEL) -> bytes:
    z = _load_zstd()
    cap = z.ZSTD_compressBound(len(data))
    dst = C.create_string_buffer(cap)
    n = z.ZSTD_compress(dst, cap, data, len(data), level)
    if z.ZSTD_isError(n):
        raise RuntimeError("ZSTD_compress failed")
    return dst.raw[:n]


def zstd_decompress(data: bytes) -> bytes:
    z = _load_zstd()
    size = z.ZSTD_getFrameContentSize(data, len(data))
    if size >= (1 << 62):
        raise ValueError("zstd frame without content size")
    dst = C.create_string_buffer(max(1, size))
    n = z.ZSTD_decompress(dst, size, data, len(data))
    if z.ZSTD_isError(n) or n != size:
        raise ValueError("ZSTD_decompress failed")
    return dst.raw[:size]


# ---- session directory / metadata (reference :43-85) ----------------------------------------
def get_recording_dir(session_name: str, root: Path = None) -> Path:
    base = Path(root or PROJECT_ROOT) / "recordings" / session_name
    base.mkdir(parents=True, exist_ok=True)
    return base


def save_metadata(rec_dir: Path, config: dict, start_time: float):
    meta = {**config, "start_time": start_time, "start_datetime": datetime.fromtimestamp(start_time).isoformat()}
    with open(Path(rec_dir) / "metadata.json", "w") as f:
        json.dump(meta, f, indent=2)


def load_metadata(rec_dir: Path) -> dict:
    with open(Path(rec_dir) / "metadata.json", "r") as f:
        return json.load(f)


def _frame_paths(rec_dir, idx):
    rec_dir = Path(rec_dir)
    return rec_dir / f"frame_{idx:04d}.zstd", rec_dir / f"frame_{idx:04d}.npz"


def get_completed_frames(rec_dir: Path) -> int:
    """Number of contiguous frames from 0 present as .npz or .zstd (reference :67-76)."""
    count = 0
    while any(p.exists() for p in _frame_paths(rec_dir, count)):
        count += 1
    return count


def find_latest_state(rec_dir: Path, max_frame: int):
    for frame in range(max_frame, -1, -1):
        p = Path(rec_dir) / f"state_{frame:04d}.npz"
        if p.exists():
            return p, frame
    return None, -1


# ---- frames (reference :88-210, :231-326) -----------------------------------------------------
def _atomically(path: Path, write):
    """`write(file object)` into a temporary name beside `path`, then os.replace: a frame or state file either exists
    complete or not at all.  An interrupt in the middle of a 24 MB frame write (a large share of a frame interval at
    1 M bodies) must not leave a truncated file that get_completed_frames() counts and a resume builds on."""
    path = Path(path)
    tmp = path.with_name("." + path.name + ".part")
    try:
        with open(tmp, "wb") as f:
            write(f)
        os.replace(tmp, path)
    except BaseException:
        try:
            tmp.unlink()
        except OSError:
            pass
        raise


def write_bytes_atomic(path: Path, blob: bytes):
    _atomically(path, lambda f: f.write(blob))


def save_frame(rec_dir: Path, frame_idx: int, positions: np.ndarray, colors: np.ndarray):
    p32, c32 = positions.astype(np.float32), colors.astype(np.float32)
    _atomically(Path(rec_dir) / f"frame_{frame_idx:04d}.npz", lambda f: np.savez(f, positions=p32, colors=c32))


def delta_quantize(cur: np.ndarray, prev: np.ndarray) -> np.ndarray:
    """int16((cur - prev) * 1000) with the reference's C-cast wrap-around beyond +-32.767
    (reference :254-262 [quirk]: lossy, wraps)."""
    with np.errstate(invalid="ignore", over="ignore"):
        return ((cur - prev) * 1000).astype(np.int16)


def pack_container(fmt: int, pos_data: bytes, col_data: bytes) -> bytes:
    """format byte + u32 length + zstd(positions payload) + u32 length + zstd(colours payload) (reference :264-279)."""
    pc, cc = zstd_compress(pos_data), zstd_compress(col_data)
    return struct.pack("B", fmt) + struct.pack("I", len(pc)) + pc + struct.pack("I", len(cc)) + cc


def compress_frame(positions, colors, prev_positions=None, prev_colors=None) -> bytes:
    use_delta = prev_positions is not None and prev_colors is not None
    if use_delta:
        return pack_container(2, delta_quantize(positions, prev_positions).tobytes(),
                              delta_quantize(colors, prev_colors).tobytes())
    return pack_container(1, positions.astype(np.float32).tobytes(), colors.astype(np.float32).tobytes())


def _split_container(data: bytes):
    if len(data) < 1:
        raise ValueError("Invalid compressed data")
    fmt = data[0]
    off = 1
    (psz,) = struct.unpack("I", data[off:off + 4])
    off += 4
    pbytes = data[off:off + psz]
    off += psz
    (csz,) = struct.unpack("I", data[off:off + 4])
    off += 4
    return fmt, pbytes, data[off:off + csz]


def decompress_frame(data: bytes, prev_positions=None, prev_colors=None):
    fmt, pbytes, cbytes = _split_container(data)
    pos_data, col_data = zstd_decompress(pbytes), zstd_decompress(cbytes)
    if fmt == 1:
        return (np.frombuffer(pos_data, dtype=np.float32).reshape(-1, 3),
                np.frombuffer(col_data, dtype=np.float32).reshape(-1, 3))
    if fmt == 2:
        if prev_positions is None or prev_colors is None:
            raise ValueError("Delta compression requires previous frame")
        dp = np.frombuffer(pos_data, dtype=np.int16).reshape(-1, 3).astype(np.float32) / 1000.0
        dc = np.frombuffer(col_data, dtype=np.int16).reshape(-1, 3).astype(np.float32) / 1000.0
        return prev_positions + dp, prev_colors + dc
    raise ValueError(f"Unknown compression format: {fmt}")


def load_frame(rec_dir: Path, frame_idx: int, prev_positions=None, prev_colors=None):
    """(positions f32 (N,3), colors f32 (N,3)); delta frames are resolved by walking back to the
    nearest absolute (.zstd format 1) or raw .npz frame, iteratively (reference :99-210)."""
    zf, nf = _frame_paths(rec_dir, frame_idx)
    if zf.exists():
        data = zf.read_bytes()
        if data and data[0] == 2 and (prev_positions is None or prev_colors is None):
            if frame_idx == 0:
                raise ValueError(f"Frame {frame_idx:04d} appears to be delta-compressed but is the first frame")
            chain = []
            k = frame_idx - 1
            base = None
            while k >= 0:
                zk, nk = _frame_paths(rec_dir, k)
                if zk.exists():
                    dk = zk.read_bytes()
                    chain.append(dk)
                    if dk[0] == 1:
                        break
                elif nk.exists():
                    with np.load(nk) as d:
                        base = (d["positions"].copy(), d["colors"].copy())
                    break
                else:
                    raise FileNotFoundError(f"Frame {k:04d} not found (needed for delta decompression)")
                k -= 1
            if base is None:
                if not chain or chain[-1][0] != 1:
                    raise ValueError(f"Frame {frame_idx:04d} is delta-compressed but no base frame found")
                base = decompress_frame(chain.pop(), None, None)
            for dk in reversed(chain):
                base = decompress_frame(dk, base[0], base[1])
            prev_positions, prev_colors = base
        return decompress_frame(data, prev_positions, prev_colors)
    if nf.exists():
        with np.load(nf) as d:
            return d["positions"].copy(), d["colors"].copy()
    raise FileNotFoundError(f"Frame {frame_idx:04d} not found")


def compress_recording(rec_dir: Path, upto: int = None, batch_size: int = COMPRESSION_BATCH_SIZE):
    """Convert raw frames to .zstd like the reference's BackgroundCompressor batches (:329-470):
    within the run every frame after the very first is a delta against its predecessor."""
    rec_dir = Path(rec_dir)
    total = get_completed_frames(rec_dir) if upto is None else upto
    prev = None
    done = 0
    for idx in range(total):
        zf, nf = _frame_paths(rec_dir, idx)
        if zf.exists():
            prev = load_frame(rec_dir, idx, *(prev or (None, None)))
            continue
        with np.load(nf) as d:
            cur = (d["positions"].copy(), d["colors"].copy())
        blob = compress_frame(cur[0], cur[1], *(prev or (None, None)))
        write_bytes_atomic(zf, blob)
        nf.unlink()
        # the decoder sees the quantised frame: chain on what it will reconstruct
        prev = decompress_frame(blob, *(prev or (None, None)))
        done += 1
    return done


# ---- initial conditions + the recording loop ---------------------------------------------------
def _generate_initial_conditions(config: dict):
    from tools.presets import generate_distribution
    p, v, m = generate_distribution(config.get("distribution", "galaxy"), config["num_bodies"],
                                    config["spawn_radius"], config["G"])
    return p.astype(np.float64), v.astype(np.float64), m.astype(np.float64)


def record(config: dict, resume: bool = False, root: Path = None, quiet: bool = False, seed=None):
    """The reference's record() on the GPU branch (:760-775, :821-876): per frame `substeps` x
    step(dt_per_frame/substeps), compute_colors(15.0), get_positions, get_colors, raw frame;
    velocities only every 50th frame for the state checkpoint.  Returns the session directory."""
    from nbody.gpu_backend import Backend, create_gpu_simulation, get_backend

    def say(*a):
        if not quiet:
            print(*a)

    rec_dir = get_recording_dir(config["session_name"], root)
    start_frame = 0
    positions = velocities = masses = None
    if resume:
        completed = get_completed_frames(rec_dir)
        if completed > 0:
            state_file, state_frame = find_latest_state(rec_dir, completed)
            if state_file is not None:
                with np.load(state_file) as st:
                    positions = st["positions"].astype(np.float64)
                    velocities = st["velocities"].astype(np.float64)
                    masses = st["masses"].astype(np.float64) if "masses" in st.files else None
                start_frame = state_frame + 1
                say(f"[Record] Resuming from frame {start_frame}")
    device_ic = bool(config.get("device_ic")) and positions is None
    if positions is None:
        # `seed` is a superset key of metadata.json: a resume / extend that finds no state checkpoint starts
        # over from frame 0 (as the reference does) and must then draw the SAME bodies (the reference, unseeded,
        # silently continues a recording with a different system)
        if seed is None:
            seed = config.get("seed")
        else:
            config = dict(config, seed=int(seed))
        if not device_ic:
            if seed is not None:
                np.random.seed(seed)
            positions, velocities, masses = _generate_initial_conditions(config)
        save_metadata(rec_dir, config, time.time())
    n = config["num_bodies"]
    total_frames = config["total_frames"]
    substeps = config["substeps"]
    dt = config["dt_per_frame"] / substeps
    if masses is None:
        masses = np.ones(n, dtype=np.float64)  # reference :752-753 [quirk]

    backend, info = get_backend()
    if backend != Backend.HIP:
        raise RuntimeError(f"[Record] no HIP backend ({info}); this build has no CPU fallback")
    if device_ic:  # extra config key: bodies drawn on the GPU (statistical parity with the presets' generator)
        from tools.presets import generate_distribution_device
        gpu_sim = generate_distribution_device(config.get("distribution", "galaxy"), n, config["spawn_radius"],
                                               config["G"], config["softening"], config["damping"],
                                               theta=config.get("theta", 0.5), seed=42 if seed is None else seed)
    else:
        gpu_sim = create_gpu_simulation(positions, velocities, masses, config["G"], config["softening"],
                                        config["damping"], theta=config.get("theta", 0.5), force_gpu=True)
    if gpu_sim is None:
        raise RuntimeError("[Record] create_gpu_simulation returned None")
    say(f"[Record] GPU acceleration: {backend.value} - {info}")
    direct_zstd = bool(config.get("zstd"))  # extra config key: write .zstd frames, delta payload quantised on the device
    if direct_zstd and start_frame > 0:
        # the delta chain continues from what a reader reconstructs for the last frame on disk
        gpu_sim.frame_set_previous(*load_frame(rec_dir, start_frame - 1))
    t0 = time.time()

    def write_frame(frame):
        gpu_sim.compute_colors(15.0)
        if direct_zstd:
            zf, _ = _frame_paths(rec_dir, frame)
            if frame == 0:
                p32, c32 = gpu_sim.frame_keyframe()
                write_bytes_atomic(zf, pack_container(1, p32.tobytes(), c32.tobytes()))
            else:
                dp, dc = gpu_sim.frame_delta()
                write_bytes_atomic(zf, pack_container(2, dp.tobytes(), dc.tobytes()))
        else:
            save_frame(rec_dir, frame, gpu_sim.get_positions(), gpu_sim.get_colors())

    def write_state(frame, compressed=False):
        x, v = gpu_sim.get_positions_f64(), gpu_sim.get_velocities()
        _atomically(rec_dir / f"state_{frame:04d}.npz",
                    lambda f: (np.savez_compressed if compressed else np.savez)(f, positions=x, velocities=v, masses=masses))

    def write_keyframe(frame):  # a frame that does not depend on the delta chain (format 1 is legal anywhere)
        gpu_sim.compute_colors(15.0)
        zf, _ = _frame_paths(rec_dir, frame)
        p32, c32 = gpu_sim.frame_keyframe()
        write_bytes_atomic(zf, pack_container(1, p32.tobytes(), c32.tobytes()))

    frame = start_frame - 1
    try:
        for frame in range(start_frame, total_frames):
            gpu_sim.step_many(dt, substeps)
            write_frame(frame)
            if (frame + 1) % STATE_EVERY == 0:
                write_state(frame)
                old = rec_dir / f"state_{frame - STATE_EVERY:04d}.npz"
                if old.exists():
                    old.unlink()
    except KeyboardInterrupt:
        # reference :916-935: "Paused at frame N" + a state file so that --resume continues from there.  Which
        # frame the DEVICE stands at is asked of the library (the interrupt is delivered when the step call
        # returns, before this loop could note anything): the checkpoint must be the state of exactly the last
        # frame on disk, or a resume would skip or repeat one frame interval.
        at = start_frame - 1 + gpu_sim.step_count() // max(substeps, 1)
        if at >= 0:
            on_disk = any(q.exists() for q in _frame_paths(rec_dir, at))
            if at == frame and not on_disk:
                # stepped, frame not written - or its write was cut short: frames reach their name only complete
                # (_atomically), so a cut-short write left nothing; the device-side delta chain may already have
                # moved on, so this frame is written absolute
                write_keyframe(at) if direct_zstd else write_frame(at)
            write_state(at, compressed=True)
        say(f"\n[Record] Paused at frame {at}; resume with record(config, resume=True)")
        gpu_sim.close()
        raise
    say(f"[Record] {total_frames - start_frame} frames in {time.time() - t0:.2f}s -> {rec_dir}")
    gpu_sim.close()
    return rec_dir


def extend_recording(session_name: str, extra_frames: int, root: Path = None, quiet: bool = True):
    """The reference's ``--extend N session`` (:1156-1199): total_frames += N in metadata.json, then resume."""
    rec_dir = get_recording_dir(session_name, root)
    if not (rec_dir / "metadata.json").exists():
        raise FileNotFoundError(f"[Record] No recording found: {session_name}")
    config = load_metadata(rec_dir)
    config["total_frames"] = int(config["total_frames"]) + int(extra_frames)
    with open(rec_dir / "metadata.json", "w") as f:
        json.dump(config, f, indent=2)
    config["session_name"] = session_name
    return record(config, resume=True, root=root, quiet=quiet)
