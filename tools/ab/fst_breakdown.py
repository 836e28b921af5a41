import sys, time, io, contextlib
sys.path.insert(0, ".")
import numpy as np, torch
from gf3_audio_modem_amd.OFDM import receiver
g = np.load("tests/golden/g6_realrec.npz"); wav = g["wav_u8"]
with contextlib.redirect_stdout(io.StringIO()):
    rx = receiver(mode="A2", encoding="XOR")
    for _ in range(3): rx.receive(wav)
eng = rx._engine(wav.dtype)
def T(fn, n=20):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t = time.perf_counter(); r = fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    return np.median(ts) * 1e3, r
t_up, x = T(lambda: eng._samples(wav))
t_sync, peaks = T(lambda: eng.sync_stream(x))
starts = (peaks + 2)[:-1]
t_mx, _ = T(lambda: int(starts.max()))
t_dem, o = T(lambda: eng.demod_frames(x, starts, want=("Hs", "He", "slope")))
def EV(fn, n=20):                                               # HIP events on the stream the kernels run on
    ts = []
    for _ in range(n + 3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return float(np.median(ts[3:]))
o1 = eng.demod_frames(x, starts, want=("Hs", "He", "slope"), split=False)
o2 = eng.demod_frames(x, starts, want=("Hs", "He", "slope"), split=True)
print("plan", eng.demod_plan(int(starts.numel())), " bits equal:", bool(torch.equal(o1["bits"], o2["bits"])),
      " Hs/He/slope equal:", all(bool(torch.equal(o1[k], o2[k])) for k in ("Hs", "He", "slope")))
print("demod_frames, event-timed: one launch %.3f ms   two-phase %.3f ms   (auto %.3f ms)" % (
    EV(lambda: eng.demod_frames(x, starts, want=("Hs", "He", "slope"), split=False)),
    EV(lambda: eng.demod_frames(x, starts, want=("Hs", "He", "slope"), split=True)),
    EV(lambda: eng.demod_frames(x, starts, want=("Hs", "He", "slope")))))
t_dec, bits = T(lambda: rx._decode_packed(eng, o["bits"]))
t_h, _ = T(lambda: (o["Hs"].cpu().numpy(), o["He"].cpu().numpy(), o["slope"].cpu().numpy()))
with contextlib.redirect_stdout(io.StringIO()):
    t_all, _ = T(lambda: rx.receive(wav))
print("upload %.3f  sync_stream %.3f  starts.max %.3f  demod_frames %.3f  unpack+xor into pinned host memory %.3f  Hs/He/slope D2H %.3f  | receive() %.3f ms" % (t_up, t_sync, t_mx, t_dem, t_dec, t_h, t_all))
