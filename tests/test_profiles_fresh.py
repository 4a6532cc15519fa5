"""tools/check_profiles.py as a test: the round's committed profile record must belong to the code in the tree -- every file under
profiles/r6/final/, every record of profiles/pmc_traffic.json and the numbers DESIGN.md section 7 quotes from them. A change to the device sources after the
record was taken turns this test red until the record is re-taken (tools/r6_profiles.sh): stale evidence fails loudly instead of being quoted."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_profiles_belong_to_this_code():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_profiles.py"), "--strict"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-4000:]


def test_the_check_notices_a_foreign_file_and_a_changed_one(tmp_path):
    """The checker itself: a manifest over three files; a fourth file dropped in afterwards and an edit to a listed one are both reported."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_profiles as cp
    d = tmp_path / "profiles" / "r6" / "final"; d.mkdir(parents=True)
    for n in ("a.txt", "b.csv", "bench.json"):
        (d / n).write_text(json.dumps({"code_hash": cp.code_hash(ROOT), "roofline": {"avg_launch_ms": 1.0, "kernel": "k"}}) if n.endswith("json") else n)
    cp.write_manifest(str(d))
    saved = cp.final_dir
    cp.final_dir = lambda rnd=cp.ROUND: str(d)
    try:
        assert not [p for p in cp.check() if p.startswith("final/")]
        (d / "stale.csv").write_text("copied from round 5"); (d / "a.txt").write_text("edited")
        problems = cp.check()
        assert any("stale.csv is not in the manifest" in p for p in problems) and any("a.txt changed" in p for p in problems)
    finally:
        cp.final_dir = saved
