"""Time the l = 55 transform with each variant library under quantum-systems_amd/variants/ (built by
tools/build_variant.sh), one child process per variant, same box.  Results of ablated builds are wrong by design.
Prints the time of the whole transform with only ONE of its two passes on the sandwich kernel, and the same
with the old path (QS_SANDWICH=0) as the reference point."""
import glob, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
l = sys.argv[1] if len(sys.argv) > 1 else "55"
def run(lib, mode):
    env = dict(os.environ, QS_SANDWICH=mode)
    if lib: env["QS_AMD_LIB"] = lib
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "small_l_profile.py"), l], env=env, capture_output=True, text=True)
    return r.stdout.strip() or r.stderr[-300:]
print("old path              :", run(None, "0"), flush=True)
for lib in sorted(glob.glob(os.path.join(root, "quantum-systems_amd", "variants", "*.so"))):
    for mode in ("2", "3", "1"):
        print(os.path.basename(lib), {"2": "(d,c) only", "3": "(b,a) only", "1": "both      "}[mode], ":", run(lib, mode), flush=True)
