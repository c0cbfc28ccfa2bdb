"""Round 6: the multi-stream codec failure as a statistics problem ("two runs per knob against a 1-3 % base rate
excludes nothing"). One trial = one fresh process of tools/race_first_round.py (the calls alone, then the FIRST concurrent round — the
only round that ever failed); an arm = a set of environment variables and tool switches. Arms run interleaved (arm 0 trial 0, arm 1
trial 0, ..) so that box drift hits all of them alike; output: failures / trials with the 95 % Wilson interval, per arm.

  python tools/race_trials.py <trials> <arm> [<arm> ..]
  arm = name:ENV=v,ENV=v,switch,switch=v     (upper-case keys are environment variables, the rest go to race_first_round.py)
  e.g. base:SSRHIP_POISON_ALLOC=1  queues:SSRHIP_POISON_ALLOC=1,prewarm-queues=8  nocache:PYTORCH_NO_CUDA_MEMORY_CACHING=1
"""
import math
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def wilson(k, n, z=1.96):
    if n == 0:
        return 0.0, 1.0
    p = k / n
    d = 1 + z * z / n
    c = p + z * z / (2 * n)
    h = z * math.sqrt(p * (1 - p) / n + z * z / (4 * n * n))
    return max(0.0, (c - h) / d), min(1.0, (c + h) / d)


def main():
    trials = int(sys.argv[1])
    arms = []
    for spec in sys.argv[2:]:
        name, _, rest = spec.partition(":")
        env, sw = {}, []
        for item in filter(None, rest.split(",")):
            key = item.split("=")[0]
            if key.isupper():
                env[key] = item.partition("=")[2]
            else:
                sw.append(item)
        arms.append((name, env, sw))
    fails = [0] * len(arms)
    done = [0] * len(arms)
    errs = [0] * len(arms)
    t0 = time.time()
    for t in range(trials):
        for i, (name, env, sw) in enumerate(arms):
            out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "race_first_round.py")] + sw, env=dict(os.environ, **env),
                                 capture_output=True, text=True, timeout=300)
            done[i] += 1
            if out.returncode == 3:
                fails[i] += 1
                keep = [ln[:400] for ln in out.stdout.splitlines() if "per-frame max" not in ln]
                print(f"[{name} trial {t}] " + "\n    ".join(keep), flush=True)
            elif out.returncode != 0:
                errs[i] += 1
                print(f"[{name} trial {t}] ERROR rc={out.returncode}: {out.stderr[-600:]}", flush=True)
        if (t + 1) % 10 == 0 or t + 1 == trials:
            print(f"--- after {t + 1} trials ({time.time() - t0:.0f} s): " +
                  "; ".join(f"{n} {f}/{d}" + (f" (+{e} errors)" if e else "") for (n, _, _), f, d, e in zip(arms, fails, done, errs)), flush=True)
    print("=== result")
    for (name, env, sw), f, d, e in zip(arms, fails, done, errs):
        lo, hi = wilson(f, d - e)
        print(f"{name:12s} {f:4d} / {d - e:4d} failed  = {100.0 * f / max(d - e, 1):5.1f} %  (95 % Wilson {100 * lo:.1f} .. {100 * hi:.1f} %)   env {env} switches {sw}" +
              (f"  [{e} trials errored]" if e else ""))


if __name__ == "__main__":
    main()
