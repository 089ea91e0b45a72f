"""The committed opcode mix that bench.py prices (profiles/*_isa_mix.json) must describe the sources the library is built
from: it is regenerated on the CPU by `python3 tools/isa_mix.py <tag>` whenever csrc/ changes."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ac-mpc_amd"))


def test_the_committed_opcode_mix_matches_the_sources():
    import bench

    mix, path = bench.newest_profile("isa_mix.json")
    assert mix is not None, "no profiles/*_isa_mix.json"
    assert mix["source_sha256"] == bench.loaded_source_hash(), "%s is stale: python3 tools/isa_mix.py" % path
    for entry in ("window_2_5", "window_1_2", "exhaustive", "fused_round"):
        assert sum(mix["entries"][entry]["valu"].values()) > 50


def test_issue_costs_price_every_class():
    import bench

    probe, _ = bench.newest_profile("valu_probe.json")
    cheap = bench.issue_cost_ns("v_add_f32_e32", probe)
    assert bench.issue_cost_ns("v_fmac_f32_e32", probe) < 1.2 * cheap
    for dear in ("v_cmp_eq_f32_e64", "v_cndmask_b32_e32", "v_min3_f32", "v_med3_i32", "v_lshl_add_u64", "v_max_f32_e32"):
        assert bench.issue_cost_ns(dear, probe) > 1.3 * cheap, dear
    assert bench.issue_cost_ns("v_some_unknown_op", probe) == cheap   # never overstates


def test_every_entry_point_is_in_the_integration_index():
    """INTEGRATION.md 3a lists every entry point of include/acmpc.h with the reference code it stands for: a new one has
    to be added there (the acmpc_pf_* family is listed by its stems)."""
    import re
    header = open(os.path.join(ROOT, "include", "acmpc.h")).read()
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    names = sorted(set(re.findall(r"\b(acmpc_[a-z_0-9]+)\s*\(", header)))
    assert len(names) > 50
    missing = []
    for name in names:
        stem = name.split("_")[-1]
        listed = name in doc or ("`_%s`" % stem in doc) or ("/ `_%s`" % stem in doc) or ("_%s`" % stem in doc)
        if not listed:
            missing.append(name)
    assert not missing, missing
