"""Randomised small cases for parity fuzzing (deterministic: seeded `random`).  Each case = (kwargs, features or
None, FASTQ bytes).  The generators deliberately produce awkward input: odd symbols in reads, libraries and anchors,
ragged and empty reads, quality lines of other lengths, bytes >= 128, CRLF, blank lines, partial records."""
import random

SEQ_ALPHA = b"ACGT" * 6 + b"acgtNnR:"
QUAL_ALPHA = b"IIIIIIII??55+#!~" + bytes([31, 127, 200])


def rand_seq(rng, n, alpha=SEQ_ALPHA):
    return bytes(rng.choice(alpha) for _ in range(n))


def make_case(seed):
    rng = random.Random(seed)
    mode = "EC" if rng.random() < 0.3 else "C"
    kw = dict(mode=mode, miss=rng.choice([0, 0, 1, 1, 2, 3]), phred=rng.choice([1, 10, 20, 30, 30, 41, 60]))
    anchored = rng.random() < 0.45
    flen = rng.choice([0, 1, 3, 6, 8, 8, 12, 20, 30, 33])
    kw["length"] = flen
    up = down = None
    if anchored:
        alen = rng.choice([1, 3, 4, 6, 9])
        clean = rng.random() < 0.8
        up = rand_seq(rng, alen, b"ACGT" if clean else b"ACGTNa").decode()
        down = rand_seq(rng, rng.choice([1, 3, 4, 6]), b"ACGT" if clean else b"ACGTn").decode()
        which = rng.choice(["both", "up", "down", "pairs"])
        if which in ("both", "up"):
            kw["upstream"] = up
        if which in ("both", "down"):
            kw["downstream"] = down
        if which == "pairs":
            kw["upstream"] = up + "," + rand_seq(rng, 4, b"ACGT").decode()
            kw["downstream"] = down + "," + rand_seq(rng, 3, b"ACGT").decode()
        kw["miss_search_up"] = rng.choice([0, 0, 1, 2])
        kw["miss_search_down"] = rng.choice([0, 0, 1, 2])
        kw["qual_up"] = rng.choice([1, 20, 30])
        kw["qual_down"] = rng.choice([1, 30, 41])
    else:
        starts = [rng.choice([0, 0, 2, 5, 17])]
        if rng.random() < 0.2:
            starts.append(rng.choice([1, 9, 14]))
        kw["start"] = ",".join(map(str, starts))
    feats = None
    if mode == "C":
        n = rng.randint(1, 25)
        alpha = b"ACGT" if rng.random() < 0.75 else b"ACGTN:"
        seen, feats = set(), []
        for _ in range(n):
            L = flen if rng.random() < 0.8 else rng.choice([0, 2, 5, 8, 17])
            if "," in kw.get("start", "") and rng.random() < 0.7:
                s = (rand_seq(rng, flen, b"ACGT") + b":" + rand_seq(rng, flen, b"ACGT")).decode()
            else:
                s = rand_seq(rng, L, alpha).decode()
            if s not in seen:
                seen.add(s)
                feats.append(s)
    # reads: many built around library features / anchors so that matches actually happen
    recs = []
    for r in range(rng.randint(0, 60)):
        rl = rng.choice([0, 3, 10, 25, 40, 40, 64, 150])
        seq = bytearray(rand_seq(rng, rl))
        if feats and rl and rng.random() < 0.7:
            f = rng.choice(feats).replace(":", "").encode()
            pos = rng.choice([0, 0, 2, 5, 17, rng.randint(0, max(0, rl - 1))])
            if anchored and up and rng.random() < 0.8:
                f = up.encode() + f + (down or "").encode()
            seq[pos:pos + len(f)] = f
            del seq[max(rl, 0) + 20:]
            if rng.random() < 0.4 and seq:
                seq[rng.randrange(len(seq))] = rng.choice(SEQ_ALPHA)
        elif anchored and up and rl > 12 and rng.random() < 0.8:
            pos = rng.randint(0, rl - 8)
            seq[pos:pos + len(up)] = up.encode()
            if down and rng.random() < 0.8:
                p2 = min(len(seq), pos + len(up) + rng.choice([0, 3, 8, 20, 35]))
                seq[p2:p2 + len(down)] = down.encode()
        ql = len(seq) if rng.random() < 0.9 else rng.choice([0, 5, len(seq) + 3])
        qual = bytes(rng.choice(QUAL_ALPHA) for _ in range(ql))
        eol = b"\r\n" if rng.random() < 0.1 else b"\n"
        recs.append(b"@r" + str(r).encode() + eol + bytes(seq) + (b"  " if rng.random() < 0.05 else b"") + eol + b"+" + eol + qual + eol)
    fq = b"".join(recs)
    t = rng.random()
    if t < 0.1:
        fq = fq[:-1] if fq else fq
    elif t < 0.2:
        fq += b"@x\nACGT\n+\n"
    elif t < 0.25:
        fq = b"\n" + fq
    # '\n' and '\r' inside quality bytes would change the framing: QUAL_ALPHA has none; 31/127/200 are legal oddities
    return kw, feats, fq
