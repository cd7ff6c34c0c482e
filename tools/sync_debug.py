"""Which calls of one training step make the HOST wait for the GPU: runs bench.py's step under torch.cuda.set_sync_debug_mode("warn") after
the warm-up steps (captures and allocations done) and prints each distinct warning with its Python stack once."""
import os, sys, warnings, traceback, runpy
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
seen = {}
orig = warnings.showwarning
def show(message, category, filename, lineno, file=None, line=None):
    if "synchroniz" in str(message).lower():
        st = "".join(traceback.format_stack(limit=14)[:-1])
        key = (filename, lineno)
        if key not in seen:
            seen[key] = (str(message), st)
    else:
        orig(message, category, filename, lineno, file, line)
warnings.showwarning = show
warnings.simplefilter("always")
sys.argv = ["bench.py", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-roofline"]
os.environ["NK_SYNC_DEBUG_AFTER_WARMUP"] = "1"
runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bench.py"), run_name="__main__")
print(f"\n{len(seen)} distinct synchronising call sites during the timed steps:")
for (fn, ln), (msg, st) in seen.items():
    print("-" * 100)
    print(msg.strip())
    print("".join(l for l in st.splitlines(True) if "/root/repo" in l or "site-packages/torch" not in l)[-1500:])
