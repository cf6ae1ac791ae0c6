"""Regenerate the row lists of hbs_scan4.hip / hbs_sparse.h for N rows per wavefront (dev aid).
usage: [CSRC=dir] set_rows4.py N [park_first]      (CSRC: a copy of hevcbitstream_amd/csrc to edit instead, for `make variant CSRC=dir`)"""
import os, re, sys
csrc = os.environ.get("CSRC", "hevcbitstream_amd/csrc")
rows = int(sys.argv[1]); park_first = int(sys.argv[2]) if len(sys.argv) > 2 else rows - 16
p = csrc + '/hbs_scan4.hip'
s = open(p).read()
s = re.sub(r'static_assert\(k4Rows == \d+, "the row lists below name every row register"\);', 'static_assert(k4Rows == %d, "the row lists below name every row register");' % rows, s)
s = re.sub(r'#define HBS_ROWS\(X\) .*', '#define HBS_ROWS(X) ' + ' '.join('X(%d)' % i for i in range(rows)), s)
s = re.sub(r'#define HBS_ROW_TRIPLES\(X\) .*', '#define HBS_ROW_TRIPLES(X) ' + ' '.join('X(%d,%d,%d)' % (i - 1, i, i + 1) for i in range(1, rows - 1)) + '   /* (previous row, row, next row), inner rows */', s)
s = re.sub(r'constexpr int kParkRows = \d+;', 'constexpr int kParkRows = %d;' % (rows - park_first), s)
s = re.sub(r'#define HBS_PARKED\(X\) .*', '#define HBS_PARKED(X) ' + ' '.join('X(%d,%d)' % (i, park_first + i) for i in range(rows - park_first)), s)
assert rows % 4 == 0, "the flag pass takes rows four at a time"


def rl(reg, comp, lane_no):
    return '(uint32_t)__builtin_amdgcn_readlane((int)R.q%d.%s, %d)' % (reg, comp, lane_no)


def group(first):
    args = []
    for r in range(first, first + 4):
        args += [str(r),
                 'R.before' if r == 0 else rl(r - 1, 'w', 63),
                 'R.after' if r == rows - 1 else rl(r + 1, 'x', 0),
                 'R.before2' if r == 0 else rl(r - 1, 'z', 63)]
    return '            HBS_FLAG_GROUP(' + ', '.join(args) + ')'


def ld4(first):
    return '            HBS_LD4(%d, %d, %d, %d)' % (first, first + 1, first + 2, first + 3)


def body():
    """the fetch and the flag pass, interleaved: group g is flagged with the rows up to 4g+7 issued, so 3 to 7 loads of the
    wavefront are in flight at any time (scripts/ubench/ceiling3.hip: deeper queues delay the other workgroup's look-back polls)"""
    ahead = int(os.environ.get("AHEAD", 1))        # groups of loads issued beyond the one a flag group's neighbour row is in (1: shipped)
    assert ahead >= 1, "a group's last row needs the first row of the next group: at least one group ahead"
    out = [ld4(4 * i) for i in range(ahead + 1) if 4 * i < rows]
    for g in range(0, rows, 4):
        out.append(group(g))
        if g + 4 * (ahead + 1) < rows:
            out.append(ld4(g + 4 * (ahead + 1)))
    return out


lines = s.split('\n')
is_gen = lambda ln: ln.startswith('            HBS_FLAG_GROUP(') or ln.startswith('            HBS_LD4(')
first = next(i for i, ln in enumerate(lines) if is_gen(ln))
last = max(i for i, ln in enumerate(lines) if is_gen(ln))
assert all(is_gen(ln) for ln in lines[first:last + 1])
lines[first:last + 1] = body()
s = '\n'.join(lines)
s = re.sub(r'static_assert\(k4Rows == \d+, "first and last row are named above"\);', 'static_assert(k4Rows == %d, "first and last row are named above");' % rows, s)
open(p, 'w').write(s)
p = csrc + '/hbs_sparse.h'
s = open(p).read()
s = re.sub(r'constexpr int k4Rows          = \d+;', 'constexpr int k4Rows          = %d;' % rows, s)
open(p, 'w').write(s)
print("rows", rows, "parked", rows - park_first)
