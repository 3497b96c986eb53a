"""Sample sclk / power with rocm-smi while forwards run back to back (is the forward power-limited?)."""
import importlib, os, subprocess, sys, threading, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import pairs
pkg = importlib.import_module("atm-vfi_amd")
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
net = pkg.NetworkBase(); net.load_state_dict(pkg.synthetic_state_dict("base", seed=1)); net.to(dev).eval()
a, b = pairs.random_pair(1, 1088, 1920, seed=3); a, b = a.to(dev), b.to(dev)
for _ in range(3): net(a, b)
torch.cuda.synchronize()
stop = False
samples = []
def watch():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "-d", "0", "--showclocks", "--showpower", "--csv"], capture_output=True, text=True, timeout=20).stdout
            samples.append((time.perf_counter(), out.strip().replace("\n", " | ")))
        except Exception as e:
            samples.append((time.perf_counter(), f"err {e}"))
        time.sleep(0.3)
print("idle:", subprocess.run(["rocm-smi", "-d", "0", "--showclocks", "--showpower", "--csv"], capture_output=True, text=True).stdout)
th = threading.Thread(target=watch); th.start()
t0 = time.perf_counter(); n = 0
while time.perf_counter() - t0 < 12:
    for _ in range(10): net(a, b)
    torch.cuda.synchronize(); n += 10
dt = time.perf_counter() - t0
stop = True; th.join()
print(f"{n} forwards, {dt / n * 1e3:.2f} ms each")
for t, s in samples: print(f"{t - t0:6.2f}s {s}")
