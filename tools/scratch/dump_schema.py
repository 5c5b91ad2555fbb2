import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
for name, typ in cur.execute("select name, type from sqlite_master where type in ('table','view')").fetchall():
    cols = [r[1] for r in cur.execute(f"pragma table_info('{name}')")]
    try:
        n = cur.execute(f"select count(*) from '{name}'").fetchone()[0]
    except Exception as e:
        n = -1
    print(typ, name, n, cols[:14])
for v in ("pmc_events", "counters_collection", "kernels"):
    try:
        for row in cur.execute(f"select * from {v} limit 2"):
            print(v, [str(x)[:40] for x in row])
    except Exception as e:
        print(v, "ERR", e)
