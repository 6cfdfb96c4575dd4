"""Deterministic generator of random price files (test DATA generator; no reference code): BOM / no BOM, quoted or not,
thousands separators, two date formats, junk lines before the header, missing and NA cells, unparsable dates, header
variants and price-column choices, ragged lines.  Shared by tests/golden/make_fuzz_goldens.py (which runs the REFERENCE's
read_csv_file on these files and stores what it returns) and tests/test_ingest_np.py."""
import datetime
import random

COLS = [["Date", "Price", "Open"], ["date", "Close", "Vol."], ["Date", "Volume", "Adj Close", "Open"], ["Time", "Date", "Value"],
        [" Date ", "PRICE"]]


def cases(n_cases=300, seed=1):
    """-> list of (text, bom)"""
    rnd = random.Random(seed)
    d0 = datetime.date(2024, 1, 1)
    out = []
    for it in range(n_cases):
        n = rnd.randint(1, 30)
        quoted, thousands = rnd.random() < 0.5, rnd.random() < 0.3
        datefmt = rnd.choice(["%m/%d/%Y", "%Y-%m-%d"])
        cols = rnd.choice(COLS)
        rows = []
        for i in range(n):
            d = d0 + datetime.timedelta(days=rnd.randint(0, 400))
            row = []
            for c in cols:
                if c.strip().lower() == "date":
                    row.append(d.strftime(datefmt) if rnd.random() > 0.05 else rnd.choice(["", "n/a", "garbage"]))
                elif c.strip().lower() == "time":
                    row.append("12:00")
                else:
                    v = rnd.choice([rnd.uniform(0.001, 5), rnd.uniform(5, 999), rnd.uniform(1000, 99999)])
                    cell = f"{v:,.{rnd.randint(0, 6)}f}" if thousands else f"{v:.{rnd.randint(0, 6)}f}"
                    row.append(rnd.choice(["", "NA", "-", "null"]) if rnd.random() < 0.05 else cell)
            rows.append(row)
        pre = [["junk", "", ""]] * rnd.choice([0, 0, 0, 1, 2])

        def fmt(r):
            return ",".join((f'"{x}"' if (quoted or "," in x) else x) for x in r)
        text = "\n".join([fmt(r[:len(cols)]) for r in pre] + [fmt(cols)] + [fmt(r) for r in rows]) + "\n"
        out.append((text, it % 2 == 0))
    return out
