"""Share of a file's substantive lines that also occur (whitespace- and comment-normalised) anywhere in the reference
tree's sources for this path.  A self-check for "written from the behaviour, not from the text"; needs /root/reference
(development container only).  usage: python tools/line_match.py <file> [<file> ...]"""
import glob
import re
import sys

REF = "/root/reference"


def norm_lines(path):
    out = []
    try:
        text = open(path, errors="ignore").read()
    except OSError:
        return out
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    for line in text.splitlines():
        line = re.sub(r"//.*", "", line)
        line = re.sub(r"\s+", "", line).replace("std::ranges::", "std::").replace("glm::", "")
        if len(line) >= 12 and not line.startswith("#include"):
            out.append(line)
    return out


def main():
    ref = set()
    for pat in ("src/audio/*.cpp", "src/audio/*.h", "tests/*.cpp", "tests/*.h", "src/mesh/Tets.cpp", "src/mesh/TetMesh.h", "src/mesh/Tetrahedralize.*", "src/numeric/Predicates.*", "src/Job.h"):
        for f in glob.glob(f"{REF}/{pat}"):
            ref.update(norm_lines(f))
    for path in sys.argv[1:]:
        mine = norm_lines(path)
        hit = sum(1 for line in mine if line in ref)
        print(f"{path}: {hit}/{len(mine)} = {100.0 * hit / max(1, len(mine)):.1f}%")


if __name__ == "__main__":
    main()
