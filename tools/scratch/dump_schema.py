import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
for name, sql in cur.execute("select name, sql from sqlite_master where type in ('table','view')").fetchall():
    try:
        n = cur.execute(f"select count(*) from '{name}'").fetchone()[0]
    except Exception as e:
        n = str(e)
    print("==", name, n)
    if n and isinstance(n, int) and ("pmc" in name.lower() or "kernel" in name.lower() or "counter" in name.lower()):
        cols = [r[1] for r in cur.execute(f"pragma table_info('{name}')")]
        print("   cols", cols)
        for row in cur.execute(f"select * from '{name}' limit 3"):
            print("   ", row)
