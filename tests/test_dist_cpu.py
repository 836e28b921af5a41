"""N>1 path on CPU: frame sharding + the single all-gather of packed bits, world size 2, gloo."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["GF3_ROOT"])
import numpy as np, torch
from gf3_audio_modem_amd import dist as gd
rank, world, local = gd.init_from_env(backend="gloo")
assert world == 2
row = 37
def rows_of(lo, hi):                      # packed bytes of frame f are a function of f only
    f = np.arange(lo, hi)[:, None]; c = np.arange(row)[None, :]
    return torch.from_numpy(((f * 131 + c * 7 + 3) % 251).astype(np.uint8))
for F_total in (8, 5, 2, 3):
    lo, hi = gd.shard_range(F_total, rank, world)
    mine = rows_of(lo, hi)
    full = gd.all_gather_bits(mine, F_total)
    assert full.shape == (F_total, row), full.shape
    assert torch.equal(full, rows_of(0, F_total)), F_total
    if F_total % world == 0:              # preallocated output, the path bench.py uses
        out = torch.empty((F_total, row), dtype=torch.uint8)
        gd.all_gather_bits(mine, out=out)
        assert torch.equal(out, rows_of(0, F_total))
# block-cyclic sharding + per-chunk gathers (the overlapped path of bench.py), CPU tensors
F_local, chunks = 12, 3
mine = gd.cyclic_frame_index(rank, world, F_local, chunks)
local = rows_of(0, world * F_local)[mine]
out = torch.empty((world * F_local, row), dtype=torch.uint8)
og = gd.OverlappedGather(out, F_local, chunks)
Fc = F_local // chunks
for c in range(chunks):
    og.chunk_done(c, local[c * Fc:(c + 1) * Fc].contiguous())
og.finish()
assert torch.equal(out, rows_of(0, world * F_local))
t = gd.max_over_ranks(float(rank + 1), torch.device("cpu"))
assert t == 2.0
gd.barrier()
import torch.distributed as dist
dist.destroy_process_group()
print("rank", rank, "ok")
'''


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def test_shard_range_partitions_every_frame_once():
    from gf3_audio_modem_amd.dist import shard_range
    for F in (0, 1, 7, 8, 65536, 1048576 + 3):
        for world in (1, 2, 3, 8):
            edges = [shard_range(F, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == F
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in edges]
            assert max(sizes) - min(sizes) <= 1


def test_all_gather_bits_two_ranks_gloo(tmp_path):
    port = _free_port()
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), GF3_ROOT=ROOT)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=180)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert f"rank {r} ok" in o


def test_cyclic_frame_index_covers_every_frame_once():
    import torch
    from gf3_audio_modem_amd.dist import cyclic_frame_index
    for world, F_local, chunks in ((1, 8, 4), (2, 12, 3), (8, 16, 4)):
        idx = torch.cat([cyclic_frame_index(r, world, F_local, chunks) for r in range(world)])
        assert sorted(idx.tolist()) == list(range(world * F_local))
        # chunk c of rank r is a contiguous block placed at c*world*Fc + r*Fc
        Fc = F_local // chunks
        assert cyclic_frame_index(1 % world, world, F_local, chunks)[Fc].item() == world * Fc + (1 % world) * Fc


def test_single_process_is_identity():
    import torch
    from gf3_audio_modem_amd import dist as gd
    x = torch.arange(12, dtype=torch.uint8).reshape(3, 4)
    assert gd.all_gather_bits(x) is x
    assert gd.max_over_ranks(1.5, torch.device("cpu")) == 1.5


def test_unpack_bits_layout_cpu():
    """Engine.unpack_bits is pure layout (torch ops): check it against np.unpackbits without a GPU."""
    import torch
    from gf3_audio_modem_amd.engine import Engine, RxConfig

    class Fake:
        cfg = RxConfig(N=1024, CP=0, P=1, D=3, data_bins=np.arange(1, 6), known_bits=np.zeros(2048, np.uint8))
    rs = np.random.RandomState(0)
    nb = Fake.cfg.bits_per_frame                       # 3*5*2 = 30 bits -> 4 bytes per frame
    bits = rs.randint(0, 2, (7, nb)).astype(np.uint8)
    packed = torch.from_numpy(np.packbits(bits, axis=1))
    got = Engine.unpack_bits(Fake, packed).numpy()
    assert np.array_equal(got, bits.reshape(-1))


STUB = r'''
import json, os, sys
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert os.environ["LOCAL_RANK"] == os.environ["RANK"] and os.environ["MASTER_ADDR"] == "127.0.0.1"
assert int(os.environ["MASTER_PORT"]) > 0
mode = sys.argv[1]
if mode == "fail" and rank == 1:
    sys.exit(7)
if mode == "hang-after-fail" and rank == 0:
    import time; time.sleep(600)
if mode == "hang-after-fail" and rank == 1:
    sys.exit(5)
if mode == "sleep":
    import time; time.sleep(600)
print("noise from rank", rank, file=sys.stderr)
if rank == 0:
    print(json.dumps({"n_gpus": world, "argv": sys.argv[1:]}))
else:
    print("rank", rank, "must not reach the parent's stdout")
'''


def test_spawn_ranks_env_rc_and_single_json_line(tmp_path):
    """The launcher behind `python bench.py --gpus N` (no torchrun): fresh child processes with torchrun's
    environment, rank 0's stdout relayed (exactly one JSON line), other ranks' stdout dropped, a failing rank's
    code propagated, and stragglers of a failed run terminated instead of waited for."""
    import json, time
    from gf3_audio_modem_amd.dist import spawn_ranks
    stub = tmp_path / "stub.py"
    stub.write_text(STUB)
    got = []
    rc, lines = spawn_ranks([sys.executable, str(stub), "ok", "--gpus", "2"], 2, relay=got.append)
    assert rc == 0 and lines == got and len(lines) == 1
    d = json.loads(lines[0])
    assert d == {"n_gpus": 2, "argv": ["ok", "--gpus", "2"]}
    rc, lines = spawn_ranks([sys.executable, str(stub), "fail"], 2, relay=got.append)
    assert rc == 7
    t0 = time.monotonic()
    rc, lines = spawn_ranks([sys.executable, str(stub), "hang-after-fail"], 2, relay=got.append, grace_s=1.0)
    assert rc == 5 and time.monotonic() - t0 < 60 and lines == []      # rank 1's own code, not rank 0's -15
    # an exception in the parent (here: raised by the relay callback) leaves no rank process behind
    import psutil
    before = {c.pid for c in psutil.Process().children(recursive=True)}

    def boom(line):
        raise KeyboardInterrupt
    with pytest.raises(KeyboardInterrupt):
        def relay_then_interrupt(line):
            got.append(line)
        # the interrupt arrives in the waiting thread: simulate it with a poll interval hook
        orig_sleep = time.sleep
        calls = {"n": 0}

        def sleep(dt):
            calls["n"] += 1
            if calls["n"] == 3:
                raise KeyboardInterrupt
            orig_sleep(dt)
        time.sleep = sleep
        try:
            spawn_ranks([sys.executable, str(stub), "sleep"], 2, relay=relay_then_interrupt)
        finally:
            time.sleep = orig_sleep
    orig = time.monotonic()
    while time.monotonic() - orig < 10:
        left = {c.pid for c in psutil.Process().children(recursive=True) if c.is_running() and c.status() != psutil.STATUS_ZOMBIE} - before
        if not left:
            break
        time.sleep(0.1)
    assert not left, left


WORKER8 = r'''
import os, sys
sys.path.insert(0, os.environ["GF3_ROOT"])
import torch
from gf3_audio_modem_amd import dist as gd
rank, world, local = gd.init_from_env(backend="gloo")
assert world == 8
F_local, chunks, row = 64, 8, 12                  # 8 ranks x 8 chunks of 8 frames
Fc = F_local // chunks
mine = gd.cyclic_frame_index(rank, world, F_local, chunks)           # global frame numbers of this rank's rows
# row of global frame g, decoded by rank r as its local frame l: tagged with all three
def tag(g, r, l):
    t = torch.zeros((len(g), row), dtype=torch.uint8)
    t[:, 0] = r
    t[:, 1] = l % 256
    t[:, 2] = (g // 256) % 256
    t[:, 3] = g % 256
    t[:, 4:] = ((g[:, None] * 7 + torch.arange(row - 4)[None, :]) % 251).to(torch.uint8)
    return t
local_rows = tag(mine, rank, torch.arange(F_local))
out = torch.empty((world * F_local, row), dtype=torch.uint8)
og = gd.OverlappedGather(out, F_local, chunks)
for c in range(chunks):
    og.chunk_done(c, local_rows[c * Fc:(c + 1) * Fc].contiguous())
og.finish()
# EVERY rank checks the FULL order: row g of the gathered array is global frame g, decoded by the rank and at the
# local position the block-cyclic map says
want = torch.empty_like(out)
for r in range(world):
    idx = gd.cyclic_frame_index(r, world, F_local, chunks)
    want[idx] = tag(idx, r, torch.arange(F_local))
assert torch.equal(out, want)
g = torch.arange(world * F_local)
assert torch.equal(out[:, 2].long() * 256 + out[:, 3].long(), g)                       # global frame order
c_of = g // (world * Fc); r_of = (g % (world * Fc)) // Fc; i_of = g % Fc
assert torch.equal(out[:, 0].long(), r_of) and torch.equal(out[:, 1].long(), c_of * Fc + i_of)
# ... and the literal single all-gather (chunks = 1) gives the contiguous-shard order
out1 = torch.empty((world * F_local, row), dtype=torch.uint8)
og1 = gd.OverlappedGather(out1, F_local, 1)
og1.chunk_done(0, local_rows)
og1.finish()
assert torch.equal(out1[rank * F_local:(rank + 1) * F_local], local_rows)
assert torch.equal(out1[:, 0].long(), torch.arange(world).repeat_interleave(F_local))
gd.barrier()
import torch.distributed as dist
dist.destroy_process_group()
if rank == 0:
    print("all 8 ranks agree")
'''


def test_overlapped_gather_world_size_8_gloo(tmp_path):
    """BASELINE config 4's collective at its real world size on CPU: 8 ranks x 8 chunks through OverlappedGather
    (one all_gather_into_tensor per chunk) started by spawn_ranks; rows are tagged with (rank, local frame, global
    frame) and EVERY rank checks the whole gathered order against cyclic_frame_index."""
    from gf3_audio_modem_amd.dist import spawn_ranks
    script = tmp_path / "worker8.py"
    script.write_text(WORKER8)
    env = dict(os.environ, GF3_ROOT=ROOT, OMP_NUM_THREADS="1", MKL_NUM_THREADS="1")
    got = []
    rc, lines = spawn_ranks([sys.executable, str(script)], 8, env=env, relay=got.append, grace_s=5.0)
    assert rc == 0, lines
    assert [l for l in lines if not l.startswith("[Gloo]")] == ["all 8 ranks agree"], lines    # (gloo announces its peers on stdout)


def test_bench_parent_launches_ranks_without_touching_the_gpu(tmp_path, monkeypatch):
    """`python bench.py --gpus 2` with WORLD_SIZE unset takes the launcher path before any device work: with the
    rank command replaced by a stub the parent relays one JSON line and exits with the ranks' code."""
    import json
    stub = tmp_path / "stub.py"
    stub.write_text(STUB)
    env = dict(os.environ, GF3_BENCH_RANK_CMD=json.dumps([sys.executable, str(stub), "ok"]))
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(out) == 1 and json.loads(out[0])["n_gpus"] == 2
    env["GF3_BENCH_RANK_CMD"] = json.dumps([sys.executable, str(stub), "fail"])
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 7
