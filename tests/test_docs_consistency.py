"""Guards against the documentation drifting from the code: every key vj_env_configure accepts is named in DESIGN.md (§7) and in the
tunables comment of include/vj.h, every flag of vj.h is exported by the Python mirror with the same value, and the numbers DESIGN.md
§4.3 quotes between its GENERATED markers are the ones of the committed bench record."""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
read = lambda *p: open(os.path.join(ROOT, *p)).read()


def test_every_configure_key_is_documented():
    src = read("clfacedetection_amd", "csrc", "vj_env.cpp")
    keys = sorted(set(re.findall(r'strcmp\(key, "([a-z0-9_]+)"\)', src)))
    assert len(keys) > 50
    design, header = read("DESIGN.md"), read("include", "vj.h")
    assert [k for k in keys if k not in design] == []
    assert [k for k in keys if k not in header] == []


def test_flags_of_the_header_and_the_python_mirror_agree():
    import clfacedetection_amd as pkg
    header = read("include", "vj.h")
    flags = dict(re.findall(r"(VJ_FLAG_[A-Z0-9_]+)\s*=\s*1u << (\d+)", header))
    assert len(flags) >= 6
    for name, bit in flags.items():
        assert getattr(pkg, name) == 1 << int(bit), name


def test_generated_numbers_of_design_are_the_bench_record():
    design = read("DESIGN.md")
    block = re.search(r"<!-- BEGIN GENERATED: numbers -->\n(.*?)<!-- END GENERATED: numbers -->", design, re.S).group(1)
    b = json.loads([l for l in open(os.path.join(ROOT, "profiles", "r04_bench.json")) if l.lstrip().startswith("{")][-1])
    assert f"{b['value'] / 1e9:.2f}e9 candidate windows/s, {b['ms_per_step']:.2f} ms per step" in block
    assert f"frac {b['roofline']['frac']:.3f}" in block
    for key in ("config4", "config5", "config5_raw_candidates"):
        e = b["extra"][key]
        assert f"**{e.get('ms_p50', e.get('ms_per_step')):.2f} ms**" in block, key
