"""Turn a rocprofv3 rocpd SQLite result (`*_results.db`) into a markdown per-kernel summary for profiles/."""
import sqlite3
import sys


def main(db, out, title):
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
    with open(out, 'w') as f:
        f.write(f"# {title}\n\nrocprofv3 --kernel-trace --stats; durations in microseconds (rocpd `top_kernels` view).\n\n")
        f.write("| kernel | calls | total_us | avg_us | % |\n|---|---|---|---|---|\n")
        for r in rows[:45]:
            f.write(f"| `{r[0][:100]}` | {r[1]} | {r[2]:.0f} | {r[3]:.2f} | {r[4]:.2f} |\n")


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2], sys.argv[3])
