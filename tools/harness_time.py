"""Wall time of the simulator harness (formula-vad_amd/simulator.py run_plan) on a synthetic plan written to a temporary directory:
N streams x S seconds of mono PCM16 WAV + label files + plan.json in the reference's schema.  python tools/harness_time.py [N=12] [S=1200]"""
import json, os, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package
pkg = load_package(); fv = pkg.binding
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
secs = int(sys.argv[2]) if len(sys.argv) > 2 else 1200
with tempfile.TemporaryDirectory() as d:
    inst = []
    base, labels = pkg.synth.make_stream(600.5, seed=900)
    base = base[0][: 600 * 48000]
    for i in range(n):
        x = np.tile(np.roll(base, 4801 * i), -(-secs // 600))[: secs * 48000]
        fv.wav_write(os.path.join(d, f"s{i}.wav"), x[None, :], 48000, pcm16=True)
        with open(os.path.join(d, f"s{i}.txt"), "w") as f:
            for a, b in labels:
                f.write(f"{a:.4f}\t{b:.4f}\tspeech\n")
        inst.append({"name": f"s{i}", "audio_path": f"s{i}.wav", "ref_path": f"s{i}.txt"})
    with open(os.path.join(d, "plan.json"), "w") as f:
        json.dump({"instances": inst, "config": {"preload_audio": True}}, f)
    ctx = fv.Context(0); ctx.load_synth(7)
    for rep in range(3):
        t0 = time.perf_counter()
        text, res = pkg.simulator.run_plan(os.path.join(d, "plan.json"), ctx=ctx, out=None)
        dt = time.perf_counter() - t0
        print(f"run {rep}: {n} streams x {secs} s: {dt:.3f} s wall (files read, GPU, host VAD, Evaluator, report), "
              f"{n * secs / dt:.0f}x realtime, {sum(len(r['segments']) for r in res)} segments", flush=True)
    ctx.close()
