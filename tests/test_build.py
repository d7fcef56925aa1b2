"""__graft_entry__.build(): what is reused is decided by content digests (lib/BUILD_MANIFEST.json), and every decision is printed."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_second_build_reuses_by_digest_and_says_so(capsys):
    import __graft_entry__ as g
    g.build()                       # whatever the tree's state was, it is built now
    capsys.readouterr()
    g.build()
    out = capsys.readouterr().out
    assert "[build] compiled" not in out and "[build] linked" not in out, out
    assert out.count("[build] up to date") == len(g.VARIANTS), out
    man = json.load(open(g.MANIFEST))
    libs = [k for k in man if k.endswith(".so")]
    assert len(libs) == len(g.VARIANTS) and all(len(v) == 64 for v in man.values())


def test_a_changed_flag_or_source_changes_the_digest(tmp_path):
    import __graft_entry__ as g
    src = tmp_path / "k.hip"
    src.write_text("// a\n")
    a = g._sha([str(src)], "hdr -O3")
    assert a == g._sha([str(src)], "hdr -O3")
    assert a != g._sha([str(src)], "hdr -O3 -DUTV2_H16=_Float16")
    src.write_text("// b\n")
    assert a != g._sha([str(src)], "hdr -O3")
