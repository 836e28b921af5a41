"""Event-timed: the frames sync of the headline batch, all-fp64 against screened (gf3_sync_frames_ex mode 1).  GF3_LIB selects the build."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import argparse, numpy as np, torch
import importlib.util
spec = importlib.util.spec_from_file_location("gf3_bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
ap = argparse.ArgumentParser(); ap.add_argument("--frames", type=int, default=65536); ap.add_argument("--stride", type=int, default=78720); ap.add_argument("--window", type=int, default=320)
args = ap.parse_args()
eng, cfg, big, payload, gaps = bench.build_workload(args, 0)
F = args.frames
exp = torch.arange(F, device="cuda", dtype=torch.int64) * args.stride + gaps + cfg.chirp_length
starts = torch.empty((F,), dtype=torch.int64, device="cuda")
work = eng.sync_frames_workspace(F)
ms64 = bench._event_ms(lambda: eng.sync_frames(big, F, args.stride, bench.WIN_LO, bench.WIN_LO + args.window, out_starts=starts))
ok64 = bool(torch.equal(starts, exp)); starts.zero_()
ms32 = bench._event_ms(lambda: eng.sync_frames(big, F, args.stride, bench.WIN_LO, bench.WIN_LO + args.window, out_starts=starts, screened=True, work=work))
print(json.dumps({"fp64_ms": ms64, "screened_ms": ms32, "fp64_exact": ok64, "screened_exact": bool(torch.equal(starts, exp)), "to_fp64": int(work[:4].view(torch.int32).item())}))
