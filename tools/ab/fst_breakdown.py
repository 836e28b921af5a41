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
t_dec, bits = T(lambda: rx._decode_device(eng.unpack_bits(o["bits"])))
t_h, _ = T(lambda: (o["Hs"].cpu().numpy(), o["He"].cpu().numpy(), o["slope"].cpu().numpy()))
with contextlib.redirect_stdout(io.StringIO()):
    t_all, _ = T(lambda: rx.receive(wav))
print("upload %.3f  sync_stream %.3f  starts.max %.3f  demod_frames %.3f  unpack+xor+D2H %.3f  Hs/He/slope D2H %.3f  | receive() %.3f ms" % (t_up, t_sync, t_mx, t_dem, t_dec, t_h, t_all))
