import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
for (name, typ) in db.execute("select name, type from sqlite_master where type in ('table','view') order by name"):
  if '_0000' in name: continue
  cols = [r[1] for r in db.execute(f"pragma table_info({name})")]
  print(typ, name, cols)
