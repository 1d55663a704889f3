"""Harvest the reference's own RNG-free known-answer values for the hot path into tests/golden/reference_known_answers.json.

The reference (ppdiffusers) cannot be imported here (PaddlePaddle is not installable), so nothing is *executed*; the values
are the literals its test-suite asserts against, read from the test sources with the file:line they come from. Run in the
build container only (it reads /root/reference); the JSON travels with the repository.

  python scripts/make_golden.py
"""
import json
import os
import re

REF = "/root/reference/ppdiffusers/tests"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                   "reference_known_answers.json")


def lines(path):
    with open(path) as f:
        return f.read().split("\n")


def sinusoid_slices():
    """tests/models/test_layers_utils.py test_sinoid_embeddings_hardcoded: three to_tensor([...]) literals"""
    path = os.path.join(REF, "models", "test_layers_utils.py")
    src = lines(path)
    start = next(i for i, l in enumerate(src) if "def test_sinoid_embeddings_hardcoded" in l)
    out = []
    for i in range(start, len(src)):
        m = re.search(r"paddle\.to_tensor\(\[([^\]]+)\]\)", src[i])
        if m:
            out.append(dict(values=[float(v) for v in m.group(1).split(",")], line=i + 1))
        if len(out) == 3:
            break
    args = [dict(downscale_freq_shift=1, flip_sin_to_cos=False), dict(downscale_freq_shift=0, flip_sin_to_cos=True), dict(scale=1000)]
    return dict(source=os.path.relpath(path, "/root/reference"), timesteps="arange(128)", embedding_dim=64,
                slice="[23:26, 47:50].flatten()", atol=0.01,
                cases=[dict(kwargs=a, **o) for a, o in zip(args, out)])


def loop_sums(fname):
    """scheduler full-loop tests: `assert abs(result_sum.item() - X) < tol` / result_mean, keyed by test name"""
    path = os.path.join(REF, "schedulers", fname)
    src = lines(path)
    out, test = {}, None
    for i, l in enumerate(src):
        m = re.match(r"\s+def (test_\w+)\(", l)
        if m:
            test = m.group(1)
        m = re.search(r"assert abs\(result_(sum|mean)\.item\(\) - ([-+0-9.eE]+)\) < ([-+0-9.eE]+)", l)
        if m and test:
            out.setdefault(test, {})[m.group(1)] = dict(value=float(m.group(2)), tol=float(m.group(3)), line=i + 1)
    return dict(source=os.path.relpath(path, "/root/reference"), tests=out)


def main():
    data = dict(note="literals asserted by the reference's own tests (no Paddle RNG involved); harvested by scripts/make_golden.py",
                sinusoid=sinusoid_slices(), ddim=loop_sums("test_scheduler_ddim.py"), euler=loop_sums("test_scheduler_euler.py"),
                pndm=loop_sums("test_scheduler_pndm.py"), dpm_multi=loop_sums("test_scheduler_dpm_multi.py"),
                lcm=loop_sums("test_scheduler_lcm.py"))
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(OUT, "w") as f:
        json.dump(data, f, indent=1, sort_keys=True)
    print(OUT, {k: (len(v.get("tests", v.get("cases", []))) if isinstance(v, dict) else v) for k, v in data.items()})


if __name__ == "__main__":
    main()
