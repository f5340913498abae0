"""The design documents cite files (profiles, tools, tests, sources) as evidence: every cited path must exist in the tree."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def cited_paths(text):
    for m in re.finditer(r"`((?:profiles|tools|tests|oracle|include|tsdf_amd)/[A-Za-z0-9_./*{},\-]+)`", text):
        path = m.group(1).rstrip(".,").split(":")[0]
        b = re.search(r"\{([^}]*)\}", path)
        if b:
            for alt in b.group(1).split(","):
                yield path[:b.start()] + alt + path[b.end():]
        else:
            yield path


def test_every_file_the_documents_cite_exists():
    missing = []
    for doc in ("DESIGN.md", "README.md", "INTEGRATION.md"):
        for path in cited_paths(open(os.path.join(ROOT, doc)).read()):
            full = os.path.join(ROOT, path)
            if not glob.glob(full) and not glob.glob(full + "*"):
                missing.append("%s cites %s" % (doc, path))
    assert not missing, "\n".join(sorted(set(missing)))
