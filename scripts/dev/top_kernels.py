import glob, sqlite3, sys
for p in glob.glob(sys.argv[1] + "/**/*.db", recursive=True):
    db = sqlite3.connect(p)
    for name, calls, total, avg, pct in db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
        print(f"  {name[:90]:90s} calls={calls} avg_us={avg:.2f}")
