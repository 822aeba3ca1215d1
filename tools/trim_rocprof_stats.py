"""Copies a rocprofv3 --stats kernel_stats.csv with kernel names cut to 100 chars
(torch's template names run to kilobytes) so the summary can be committed under
profiles/."""
import csv
import sys

src, dst = sys.argv[1], sys.argv[2]
with open(src) as f, open(dst, "w", newline="") as g:
  r = csv.reader(f)
  w = csv.writer(g)
  for row in r:
    row[0] = row[0][:100]
    w.writerow(row)
