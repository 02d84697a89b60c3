"""Per-kernel statistics and a stretch of the timeline out of a rocprofv3 results database (the default output format):
python3 scripts/trace_timeline.py <results.db> [kernel-name substring to centre the stretch on] [which eighth]"""
import collections, re, sqlite3, statistics, sys
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end from kernels order by start"))
short = lambda n: re.sub(r"<.*", "", re.sub(r"^void ", "", n)).replace("mhip::", "").replace("(anonymous namespace)::", "").split("(")[0]
st = collections.defaultdict(list)
for n, s, e in rows:
    st[short(n)].append(e - s)
for k, v in sorted(st.items(), key=lambda kv: -sum(kv[1]))[:16]:
    print("%-36s %6d  tot %8.2f ms  avg %7.2f us  med %7.2f" % (k[:36], len(v), sum(v) / 1e6, sum(v) / len(v) / 1e3, statistics.median(v) / 1e3))
key = sys.argv[2] if len(sys.argv) > 2 else "k_body"
part = int(sys.argv[3]) if len(sys.argv) > 3 else 1
idx = [i for i, r in enumerate(rows) if key in r[0]]
mid = idx[len(idx) * part // 8]
for i in range(mid, min(len(rows), mid + 22)):
    n, s, e = rows[i]
    print("%-36s dur %6.2f us  start-to-start %6.2f us" % (short(n)[:36], (e - s) / 1e3, (s - rows[i - 1][1]) / 1e3))
