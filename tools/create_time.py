"""vb2_ctx_create on a C3-shaped sample: wall-clock and HOST CPU time of the calling thread (+ the process, which counts
helper threads) per create, for the 42-code and the 118-code alphabet and the three flatten modes.
  python tools/create_time.py [markers]          (rocprofv3 --kernel-trace --stats on it gives the device side)"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import verifybamid_amd as vb
from verifybamid_amd import _abi

M = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
out = {}
for name, (qlo, qhi) in (("q20_40", (20, 40)), ("q2_60", (2, 60))):
    d = vb.synth.make_pileup(M, 30, 4, 0.05, 2, q_lo=qlo, q_hi=qhi)
    for mode, (hf, hp) in (("device", (0, 0)), ("host_classify", (1, 0)), ("host", (1, 1))):
        _abi.set_tunable("host_flatten", hf)
        _abi.set_tunable("host_pack", hp)
        _abi.set_tunable("flatten_threads", 1)
        vb.LikelihoodContext(d).close()
        wall, cpu_t, cpu_p = [], [], []
        for _ in range(7):
            t0, c0, p0 = time.perf_counter(), time.thread_time(), time.process_time()
            c = vb.LikelihoodContext(d)
            wall.append(time.perf_counter() - t0); cpu_t.append(time.thread_time() - c0); cpu_p.append(time.process_time() - p0)
            c.close()
        med = lambda v: 1e3 * sorted(v)[len(v) // 2]
        out["%s.%s" % (name, mode)] = {"wall_ms": round(med(wall), 3), "thread_cpu_ms": round(med(cpu_t), 3),
                                       "process_cpu_ms": round(med(cpu_p), 3)}
        print("%-8s %-14s wall %.2f ms, calling thread's CPU %.2f ms, process CPU %.2f ms" %
              (name, mode, med(wall), med(cpu_t), med(cpu_p)), flush=True)
print(json.dumps(out))
