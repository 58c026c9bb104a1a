"""developer aid: force the in-process build (hiprtc, no device needed) of the wide kernels of sum-of-Gaussians formulas with
p = 12, 32, 62 parameters -- compile errors of the kernel templates show up here, on the CPU, before a GPU minute is spent;
prints the build time per formula (analytic + forward-difference units) and the register / LDS use of the pass kernels"""
import sys, os, time, ctypes as C, tempfile, subprocess, glob
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
cache = tempfile.mkdtemp(prefix="gslnls_rtc_")
os.environ["GSLNLS_JIT_CACHE"] = cache
from gslnls_amd import _lib
L = _lib.lib()
for ng in (int(a) for a in (sys.argv[1:] or ["4", "10", "20"])):
    names, terms = [], []
    for k in range(1, ng + 1):
        names += ["a%d" % k, "m%d" % k, "s%d" % k]
        terms.append("a%d*exp(-(x-m%d)^2/s%d^2)" % (k, k, k))
    names += ["c0", "c1"]
    rhs = " + ".join(terms) + " + c0 + c1*x"
    m = _lib.Model(_lib.MODEL_EXPR, len(names), 1, None, 0)
    keep = _lib.set_expr(m, rhs, names, ["x"], "jit")
    buf = C.create_string_buffer(512)
    t0 = time.perf_counter()
    rc = L.gslnls_expr_build(C.byref(m), buf, 512)
    print("p = %d: rc %d, %.2f s, %s" % (len(names), rc, time.perf_counter() - t0, buf.value.decode()))
    if rc == 0 and os.path.exists("/opt/rocm/lib/llvm/bin/llvm-readelf"):
        for f in sorted(glob.glob(os.path.join(cache, "*"))):
            out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", f], capture_output=True, text=True).stdout
            name = None
            for line in out.splitlines():
                line = line.strip()
                if line.startswith(".name:") and "wide_" in line:
                    name = line.split()[-1][:60]
                if name and (line.startswith(".vgpr_count") or line.startswith(".sgpr_count") or line.startswith(".group_segment_fixed_size") or line.startswith(".private_segment_fixed_size") or line.startswith(".agpr_count")):
                    print("     ", os.path.basename(f)[:12], name, line)
        for f in glob.glob(os.path.join(cache, "*")):
            os.remove(f)
