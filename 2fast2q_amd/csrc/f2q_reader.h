// File side of reads_counter (fast2q.py:560-578: `gzip.open(raw)` / `open(raw)` by extension, then line iteration).
// Host only (no HIP): a byte source that hands out the decoded FASTQ text in caller-sized pieces.
//
//   PLAIN  regular files are read with several pread() workers straight into the caller's (pinned) buffer
//   BGZF   blocked gzip (bgzip, BCL Convert, samtools): every member carries its compressed size in a 'BC'
//          extra field, so the members of one batch are inflated by a pool of workers, each straight into its
//          final place in the caller's buffer; CRC32 and ISIZE of every member are checked like gzip does
//   GZIP   anything else that starts with 1f 8b.  Regular files are memory-mapped and a member of 8 MiB or more is
//          decoded by the worker pool in parallel (f2q_pargz.h: block starts are searched for, chunks are decoded
//          with an unknown window and resolved in order); small members, pipes and F2Q_GZ_PAR=0 take one sequential
//          raw inflate on the reader thread.  The CRC32 of the text, which gzread would compute on the same thread,
//          is taken per piece by the worker pool (slices + crc32_combine); header/trailer/multi-member handling is
//          done here
//
// A BGZF file that turns into ordinary gzip half way (concatenated files) is continued as GZIP from that member.  A damaged or cut-off stream delivers the text before the damage and then
// reports `truncated()` — the reference keeps what it counted before the damage and warns (:405-407), the harness does the same.
#pragma once
#include <fcntl.h>
#include <stdint.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <string>
#include <thread>
#include <vector>

#include "f2q_inflate.h"
#include "f2q_pargz.h"

struct TextSource {
    enum Kind { NONE, PLAIN, GZIP, BGZF };
    Kind kind = NONE;
    int fd = -1;
    // GZIP state: buffered compressed input, one raw inflate stream per member
    std::vector<uint8_t> zin; size_t zpos = 0, zlen = 0; bool zeof = false;
    z_stream zs; bool zs_live = false, in_member = false;
    uint32_t m_crc = 0; uint64_t m_len = 0;      // CRC32 / length of the current member's text so far
    // regular files: the compressed bytes are memory-mapped and decoded by f2qz::Inflater (F2Q_ZLIB=1: zlib instead)
    const uint8_t *zmap = nullptr; size_t zmap_len = 0, zoff = 0;
    f2qz::Inflater *infl = nullptr;
    f2qz::ParGunzip *par = nullptr; bool par_member = false, par_allowed = true;     // the current member is decoded by the pool
    size_t par_min = (size_t)8 << 20;            // ... when at least this much compressed data lies ahead (F2Q_GZ_PAR_MIN_KB)
    std::vector<f2qz::Inflater *> worker_infl;   // BGZF: one decoder (55 KB of tables) per worker of the pool
    bool use_zlib = false;
    std::string path;
    bool regular = false, bad = false, done = false;
    uint64_t file_size = 0, file_pos = 0;       // PLAIN / BGZF: compressed bytes taken from the file
    uint64_t out_total = 0;                     // decoded bytes handed out so far
    int n_threads = 1;
    // BGZF state
    std::vector<uint8_t> cbuf;                  // compressed bytes not yet inflated
    size_t cpos = 0;                            // first unconsumed byte of cbuf (mapped files: of the file)
    const uint8_t *bmap = nullptr; size_t bmap_len = 0;   // the whole file, memory-mapped: no read-ahead copy, the workers fault their own pages in
    bool c_eof = false;
    std::vector<uint8_t> spill;                 // a member larger than the caller's remaining room
    size_t spill_pos = 0;

    ~TextSource() { close(); }

    static int default_threads()
    {
        const char *e = getenv("F2Q_IO_THREADS");
        if (e && atoi(e) >= 1) return std::min(atoi(e), 64);
        unsigned hw = std::thread::hardware_concurrency();
        return (int)std::max(1u, std::min(hw ? hw : 1u, 16u));
    }

    bool truncated() const { return bad; }
    const char *kind_name() const { return kind == PLAIN ? "plain" : kind == BGZF ? "bgzf" : kind == GZIP ? "gzip" : "none"; }

    // 0 on success
    int open(const char *p, std::string &err)
    {
        close();
        path = p; n_threads = default_threads();
        { const char *z = getenv("F2Q_ZLIB"); use_zlib = z && z[0] == '1'; }
        { const char *z = getenv("F2Q_GZ_PAR"); par_allowed = !(z && z[0] == '0'); }
        { const char *z = getenv("F2Q_GZ_PAR_MIN_KB"); if (z && atol(z) >= 1) par_min = (size_t)atol(z) << 10; }
        fd = ::open(p, O_RDONLY);
        if (fd < 0) { err = std::string("cannot open ") + p; return -1; }
        struct stat st;
        regular = fstat(fd, &st) == 0 && S_ISREG(st.st_mode);
        file_size = regular ? (uint64_t)st.st_size : 0;
        uint8_t head[4096];
        size_t got = 0;
        if (regular) { ssize_t r = pread(fd, head, sizeof head, 0); got = r > 0 ? (size_t)r : 0; }
        if (got >= 2 && head[0] == 0x1f && head[1] == 0x8b) {
            uint32_t bsize;
            if (bgzf_header(head, got, bsize)) {
                kind = BGZF;
                const char *nm = getenv("F2Q_NO_MMAP");
                if (file_size > 0 && !(nm && nm[0] == '1')) {
                    void *m = mmap(nullptr, (size_t)file_size, PROT_READ, MAP_PRIVATE, fd, 0);
                    if (m != MAP_FAILED) { bmap = (const uint8_t *)m; bmap_len = (size_t)file_size; file_pos = file_size; c_eof = true; }
                }
                return 0;
            }
            return open_gzip(0, err);
        }
        if (!regular) {                                   // pipes cannot be sniffed: go by the name, like upstream (:567)
            const size_t L = path.size();
            if (L >= 3 && path.compare(L - 3, 3, ".gz") == 0) return open_gzip(0, err);
        }
        kind = PLAIN;
        return 0;
    }

    void close()
    {
        if (zs_live) { inflateEnd(&zs); zs_live = false; }
        if (zmap) { munmap(const_cast<uint8_t *>(zmap), zmap_len); zmap = nullptr; zmap_len = 0; }
        if (bmap) { munmap(const_cast<uint8_t *>(bmap), bmap_len); bmap = nullptr; bmap_len = 0; }
        delete infl; infl = nullptr; zoff = 0;
        delete par; par = nullptr; par_member = false;
        for (auto *w : worker_infl) delete w;
        worker_infl.clear();
        in_member = false; zpos = zlen = 0; zeof = false; m_crc = 0; m_len = 0;
        if (fd >= 0) { ::close(fd); fd = -1; }
        kind = NONE; bad = done = c_eof = false; file_pos = out_total = 0; cbuf.clear(); cpos = 0; spill.clear(); spill_pos = 0;
    }

    // up to `cap` bytes of text into dst; 0 = end of the data (then see truncated())
    size_t read(uint8_t *dst, size_t cap)
    {
        if (done || cap == 0) return 0;
        size_t n = 0;
        if (kind == PLAIN) n = read_plain(dst, cap);
        else if (kind == GZIP) n = zmap ? read_gzip_mm(dst, cap) : read_gzip(dst, cap);
        else if (kind == BGZF) n = read_bgzf(dst, cap);
        if (n == 0) done = true;
        out_total += n;
        return n;
    }

    // plain regular files: up to `cap` bytes starting at byte `off` (parallel pread); short only at the end of the file
    size_t read_at(uint64_t off, uint8_t *dst, size_t cap)
    {
        if (kind != PLAIN || !regular) return 0;
        file_pos = off;
        return read_plain(dst, cap);
    }

    // BGZF regular files: start over at the member that begins at compressed offset `off`
    bool seek_bgzf(uint64_t off)
    {
        if (kind != BGZF || !regular) return false;
        if (bmap) { cpos = (size_t)std::min<uint64_t>(off, bmap_len); file_pos = file_size; c_eof = true; }
        else { file_pos = off; cbuf.clear(); cpos = 0; c_eof = false; }
        spill.clear(); spill_pos = 0; bad = false; done = false;
        return true;
    }
    // BGZF regular files: compressed offset and text bytes of every member, in file order.  false: some member is not
    // a BGZF member (ordinary gzip appended, damage) -- such a file is read front to back only.
    bool bgzf_index(std::vector<uint64_t> &off, std::vector<uint32_t> &isz)
    {
        off.clear(); isz.clear();
        if (kind != BGZF || !regular) return false;
        const size_t CH = (size_t)8 << 20;
        std::vector<uint8_t> b(CH + 65536 + 64);
        uint64_t pos = 0;                                   // compressed offset of the next member
        while (pos < file_size) {
            const size_t want = (size_t)std::min<uint64_t>(b.size(), file_size - pos);
            size_t n = 0;
            while (n < want) { ssize_t r = pread(fd, b.data() + n, want - n, (off_t)(pos + n)); if (r <= 0) return false; n += (size_t)r; }
            size_t o = 0;
            while (o < n) {
                uint32_t bsize;
                if (n - o < 18 || !bgzf_header(b.data() + o, n - o, bsize)) { if (n - o >= 18 || pos + n >= file_size) return false; break; }
                if (n - o < bsize) { if (pos + n >= file_size) return false; break; }      // member continues in the next chunk
                const uint8_t *t = b.data() + o + bsize - 4;
                off.push_back(pos + o); isz.push_back(t[0] | (t[1] << 8) | (t[2] << 16) | ((uint32_t)t[3] << 24));
                o += bsize;
                if (o >= CH) break;
            }
            if (o == 0) return false;
            pos += o;
        }
        return true;
    }

private:
    // continue (or start) as ordinary gzip from compressed offset `from`
    int open_gzip(uint64_t from, std::string &err)
    {
        (void)err;
        kind = GZIP; file_pos = from; in_member = false;
        if (regular && !use_zlib && file_size > 0) {
            void *m = mmap(nullptr, (size_t)file_size, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m != MAP_FAILED) {
                (void)madvise(m, (size_t)file_size, MADV_SEQUENTIAL);
                zmap = (const uint8_t *)m; zmap_len = (size_t)file_size; zoff = (size_t)from;
                infl = new f2qz::Inflater();
                return 0;
            }
        }
        zin.resize((size_t)4 << 20); zpos = zlen = 0; zeof = false;
        return 0;
    }

    // ---- plain --------------------------------------------------------------------------------------
    size_t read_plain(uint8_t *dst, size_t cap)
    {
        if (!regular) {                                   // pipes, /dev/stdin: sequential
            size_t n = 0;
            while (n < cap) {
                ssize_t r = ::read(fd, dst + n, cap - n);
                if (r <= 0) break;
                n += (size_t)r;
            }
            return n;
        }
        const uint64_t left = file_size > file_pos ? file_size - file_pos : 0;
        const size_t want = (size_t)std::min<uint64_t>(left, cap);
        if (want == 0) return 0;
        const size_t slice_min = (size_t)4 << 20;
        const int T = (int)std::max<size_t>(1, std::min<size_t>((size_t)n_threads, want / slice_min));
        std::atomic<bool> short_read{false};
        auto work = [&](int t) {
            const size_t a = want * (size_t)t / (size_t)T, b = want * (size_t)(t + 1) / (size_t)T;
            size_t o = a;
            while (o < b) {
                ssize_t r = pread(fd, dst + o, b - o, (off_t)(file_pos + o));
                if (r <= 0) { short_read = true; return; }
                o += (size_t)r;
            }
        };
        if (T == 1) work(0);
        else {
            std::vector<std::thread> th;
            for (int t = 1; t < T; t++) th.emplace_back(work, t);
            work(0);
            for (auto &x : th) x.join();
        }
        if (short_read) { bad = true; return 0; }         // the file shrank under us
        file_pos += want;
        return want;
    }

    // ---- ordinary gzip -----------------------------------------------------------------------------
    bool z_fill()
    {
        if (zeof) return false;
        zpos = zlen = 0;
        while (zlen < zin.size()) {
            ssize_t r = regular ? pread(fd, zin.data() + zlen, zin.size() - zlen, (off_t)file_pos) : ::read(fd, zin.data() + zlen, zin.size() - zlen);
            if (r <= 0) { zeof = true; break; }
            zlen += (size_t)r; file_pos += (uint64_t)r;
            if (!regular) break;                          // pipes: take what is there
        }
        return zlen > 0;
    }
    int z_get() { if (zpos == zlen && !z_fill()) return -1; return zin[zpos++]; }
    bool z_skip(size_t n) { while (n--) if (z_get() < 0) return false; return true; }
    bool z_skip_zstring() { for (;;) { int c = z_get(); if (c < 0) return false; if (c == 0) return true; } }

    // 1 = a member header was read, 0 = clean end of the data (or trailing garbage, ignored like gzread does), -1 = cut off
    int z_member_header()
    {
        int c0 = z_get();
        while (c0 == 0) c0 = z_get();                     // zero padding between / after members
        if (c0 < 0) return 0;
        const int c1 = z_get();
        if (c0 != 0x1f || c1 != 0x8b) return 0;
        const int cm = z_get(), flg = z_get();
        if (cm != 8 || flg < 0 || (flg & 0xE0)) return -1;
        if (!z_skip(6)) return -1;                        // mtime, xfl, os
        if (flg & 4) { const int a = z_get(), b = z_get(); if (b < 0 || !z_skip((size_t)a | ((size_t)b << 8))) return -1; }
        if ((flg & 8) && !z_skip_zstring()) return -1;
        if ((flg & 16) && !z_skip_zstring()) return -1;
        if ((flg & 2) && !z_skip(2)) return -1;
        return 1;
    }

    static uint32_t crc_parallel(const uint8_t *p, size_t n, int threads)
    {
        const size_t slice_min = (size_t)2 << 20;
        (void)f2qz::Crc32::get();                          // tables built before any worker uses them
        const int T = (int)std::max<size_t>(1, std::min<size_t>((size_t)threads, n / slice_min));
        if (T == 1) return f2qz::Crc32::get().update(0u, p, n);
        std::vector<uint32_t> part((size_t)T);
        auto work = [&](int t) {
            const size_t a = n * (size_t)t / (size_t)T, b = n * (size_t)(t + 1) / (size_t)T;
            part[(size_t)t] = f2qz::Crc32::get().update(0u, p + a, b - a);
        };
        std::vector<std::thread> th;
        for (int t = 1; t < T; t++) th.emplace_back(work, t);
        work(0);
        for (auto &x : th) x.join();
        uLong c = part[0];
        for (int t = 1; t < T; t++) c = crc32_combine(c, part[(size_t)t], (z_off_t)(n * (size_t)(t + 1) / (size_t)T - n * (size_t)t / (size_t)T));
        return (uint32_t)c;
    }
    void z_account(const uint8_t *p, size_t n)
    {
        if (!n) return;
        const uint32_t c = crc_parallel(p, n, n_threads);
        m_crc = m_len ? (uint32_t)crc32_combine(m_crc, c, (z_off_t)n) : c;
        m_len += n;
    }

    size_t read_gzip(uint8_t *dst, size_t cap)
    {
        size_t n = 0, seg = 0;                            // seg: start of the current member's text inside dst
        while (n < cap && !bad) {
            if (!in_member) {
                const int h = z_member_header();
                if (h < 0) { bad = true; break; }
                if (h == 0) break;
                memset(&zs, 0, sizeof zs);
                if (inflateInit2(&zs, -15) != Z_OK) { bad = true; break; }
                zs_live = true; in_member = true; m_crc = 0; m_len = 0; seg = n;
            }
            if (zpos == zlen && !z_fill()) { bad = true; break; }         // member cut off
            zs.next_in = zin.data() + zpos; zs.avail_in = (uInt)std::min<size_t>(zlen - zpos, 1u << 30);
            zs.next_out = dst + n; zs.avail_out = (uInt)std::min<size_t>(cap - n, 1u << 30);
            const uInt in0 = zs.avail_in, out0 = zs.avail_out;
            const int r = inflate(&zs, Z_NO_FLUSH);
            zpos += in0 - zs.avail_in; n += out0 - zs.avail_out;
            if (r == Z_STREAM_END) {
                inflateEnd(&zs); zs_live = false; in_member = false;
                z_account(dst + seg, n - seg);
                uint32_t t[2] = {0, 0};
                bool ok = true;
                for (int k = 0; k < 8 && ok; k++) { const int c = z_get(); if (c < 0) ok = false; else t[k >> 2] |= (uint32_t)c << (8 * (k & 3)); }
                if (!ok || t[0] != m_crc || t[1] != (uint32_t)m_len) { bad = true; break; }
            } else if (r != Z_OK && r != Z_BUF_ERROR) { bad = true; break; }
            else if (r == Z_BUF_ERROR && zs.avail_in != 0 && zs.avail_out != 0) { bad = true; break; }
        }
        if (in_member) z_account(dst + seg, n - seg);     // the member goes on in the next piece
        return n;
    }

    // gzip member header at zmap[zoff..): same verdicts as z_member_header
    int mm_member_header()
    {
        while (zoff < zmap_len && zmap[zoff] == 0) zoff++;
        if (zoff >= zmap_len) return 0;
        if (zmap_len - zoff < 2 || zmap[zoff] != 0x1f || zmap[zoff + 1] != 0x8b) return 0;
        size_t p = zoff + 2;
        auto get = [&]() -> int { return p < zmap_len ? zmap[p++] : -1; };
        const int cm = get(), flg = get();
        if (cm != 8 || flg < 0 || (flg & 0xE0)) return -1;
        for (int k = 0; k < 6; k++) if (get() < 0) return -1;
        if (flg & 4) { const int a = get(), b = get(); if (b < 0) return -1; const size_t n = (size_t)a | ((size_t)b << 8); if (zmap_len - p < n) return -1; p += n; }
        if (flg & 8) { for (;;) { const int c = get(); if (c < 0) return -1; if (c == 0) break; } }
        if (flg & 16) { for (;;) { const int c = get(); if (c < 0) return -1; if (c == 0) break; } }
        if (flg & 2) { if (get() < 0 || get() < 0) return -1; }
        zoff = p;
        return 1;
    }

    size_t read_gzip_mm(uint8_t *dst, size_t cap)
    {
        size_t n = 0, seg = 0;
        while (n < cap && !bad) {
            if (!in_member) {
                const int h = mm_member_header();
                if (h < 0) { bad = true; break; }
                if (h == 0) break;
                // a member of 8 MiB or more (of the file: a member does not say how long it is) goes to the worker pool
                par_member = par_allowed && n_threads >= 2 && zmap_len - zoff >= par_min;
                if (par_member) { if (!par) par = new f2qz::ParGunzip(); par->start(zmap + zoff, zmap_len - zoff, n_threads); }
                else infl->reset(zmap + zoff, zmap_len - zoff);
                in_member = true; m_crc = 0; m_len = 0; seg = n;
            }
            size_t got = 0;
            const f2qz::Inflater::Status r = par_member ? par->read(dst + n, cap - n, &got) : infl->run(dst + n, dst + cap, &got);
            n += got;
            if (r == f2qz::Inflater::ERR) { bad = true; break; }
            if (r == f2qz::Inflater::OUT_FULL && par_member) break;                // (the pool may end a piece early: its next round would not fit)
            if (r == f2qz::Inflater::DONE) {
                in_member = false;
                if (par_member) { m_crc = par->crc; m_len = par->total; }          // (the pool took the CRC of what it decoded)
                else z_account(dst + seg, n - seg);
                zoff = (size_t)((par_member ? par->input_end() : infl->input_pos()) - zmap);
                if (zmap_len - zoff < 8) { bad = true; break; }
                const uint8_t *t = zmap + zoff;
                const uint32_t crc = t[0] | (t[1] << 8) | (t[2] << 16) | ((uint32_t)t[3] << 24);
                const uint32_t isz = t[4] | (t[5] << 8) | (t[6] << 16) | ((uint32_t)t[7] << 24);
                zoff += 8;
                if (crc != m_crc || isz != (uint32_t)m_len) { bad = true; break; }
            }
        }
        if (in_member && !par_member) z_account(dst + seg, n - seg);
        return n;
    }

    // ---- BGZF ---------------------------------------------------------------------------------------
    // gzip member header with a 'BC' extra subfield: *bsize = whole member size in bytes
    static bool bgzf_header(const uint8_t *h, size_t avail, uint32_t &bsize)
    {
        if (avail < 18 || h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || h[3] != 4) return false;     // BGZF: FLG = FEXTRA only
        const uint32_t xlen = h[10] | (h[11] << 8);
        if (avail < 12 + (size_t)xlen) return false;
        for (uint32_t o = 0; o + 4 <= xlen;) {
            const uint8_t *sf = h + 12 + o;
            const uint32_t slen = sf[2] | (sf[3] << 8);
            if (sf[0] == 'B' && sf[1] == 'C' && slen == 2 && o + 6 <= xlen) { bsize = (sf[4] | (sf[5] << 8)) + 1u; return true; }
            o += 4 + slen;
        }
        return false;
    }

    struct Member { size_t c_off; uint32_t c_len, hdr, isize; size_t o_off; };

    bool fill_cbuf(size_t want_more)
    {
        if (c_eof || bmap) return false;
        const size_t old = cbuf.size();
        cbuf.resize(old + want_more);
        size_t n = 0;
        while (n < want_more) {
            ssize_t r = pread(fd, cbuf.data() + old + n, want_more - n, (off_t)file_pos);
            if (r <= 0) { c_eof = true; break; }
            n += (size_t)r; file_pos += (uint64_t)r;
        }
        cbuf.resize(old + n);
        return n > 0;
    }

    f2qz::Inflater *worker_decoder(int t)
    {
        if (use_zlib) return nullptr;
        while ((int)worker_infl.size() <= t) worker_infl.push_back(new f2qz::Inflater());
        return worker_infl[(size_t)t];
    }

    static bool inflate_member(const uint8_t *src, const Member &m, uint8_t *out, f2qz::Inflater *inf)
    {
        if (m.c_len < m.hdr + 8) return false;
        const uint8_t *tail = src + m.c_off + m.c_len - 8;
        const uint32_t crc = tail[0] | (tail[1] << 8) | (tail[2] << 16) | ((uint32_t)tail[3] << 24);
        if (!inf) {                                               // F2Q_ZLIB=1
            z_stream zs; memset(&zs, 0, sizeof zs);
            if (inflateInit2(&zs, -15) != Z_OK) return false;
            zs.next_in = const_cast<Bytef *>(src + m.c_off + m.hdr); zs.avail_in = m.c_len - m.hdr - 8;
            zs.next_out = out; zs.avail_out = m.isize;
            Bytef dummy[8];
            if (m.isize == 0) { zs.next_out = dummy; zs.avail_out = sizeof dummy; }
            const int r = inflate(&zs, Z_FINISH);
            const bool ok = (r == Z_STREAM_END) && zs.total_out == m.isize && zs.avail_in == 0;
            inflateEnd(&zs);
            if (!ok) return false;
        } else {
            inf->reset(src + m.c_off + m.hdr, m.c_len - m.hdr - 8);
            size_t got = 0;
            uint8_t none[1];
            const f2qz::Inflater::Status r = inf->run(m.isize ? out : none, (m.isize ? out : none) + m.isize, &got);
            if (r != f2qz::Inflater::DONE || got != m.isize || inf->input_pos() != tail) return false;
        }
        return f2qz::Crc32::get().update(0u, out, m.isize) == crc;
    }

    size_t read_bgzf(uint8_t *dst, size_t cap)
    {
        size_t n = 0;
        if (spill_pos < spill.size()) {                   // rest of a member that did not fit last time
            const size_t k = std::min(cap, spill.size() - spill_pos);
            memcpy(dst, spill.data() + spill_pos, k);
            spill_pos += k; n = k;
            if (spill_pos == spill.size()) { spill.clear(); spill_pos = 0; }
            return n;
        }
        if (bad) return 0;
        const bool mapped = bmap != nullptr;
        auto CD = [&]() -> const uint8_t * { return mapped ? bmap : cbuf.data(); };     // the compressed bytes at hand (a mapped file: all of it)
        auto CN = [&]() -> size_t { return mapped ? bmap_len : cbuf.size(); };
        if (!mapped && cpos > 0) { cbuf.erase(cbuf.begin(), cbuf.begin() + (ptrdiff_t)cpos); cpos = 0; }   // offsets below index cbuf: compact only here
        std::vector<Member> ms;
        size_t scan = cpos, o_off = 0;
        bool foreign = false;
        for (;;) {
            if (CN() - scan < 18) {
                if (!fill_cbuf((size_t)32 << 20)) {
                    if (CN() - scan > 0) bad = true;       // a few stray bytes: cut-off header
                    break;
                }
                continue;
            }
            uint32_t bsize;
            const uint8_t *h = CD() + scan;
            const uint32_t xlen = h[10] | (h[11] << 8);
            if (CN() - scan < 12 + (size_t)xlen && !c_eof) { fill_cbuf((size_t)32 << 20); continue; }
            if (!bgzf_header(h, CN() - scan, bsize)) {
                if (h[0] == 0x1f && h[1] == 0x8b) foreign = true; else bad = true;     // ordinary gzip member / garbage
                break;
            }
            if (CN() - scan < bsize) {
                if (!fill_cbuf((size_t)32 << 20)) { bad = true; break; }               // member cut off
                continue;
            }
            if (bsize < 12 + xlen + 8) { bad = true; break; }
            const uint8_t *tail = CD() + scan + bsize - 4;
            const uint32_t isize = tail[0] | (tail[1] << 8) | (tail[2] << 16) | ((uint32_t)tail[3] << 24);
            if (o_off + isize > cap) {
                if (!ms.empty()) break;                           // next call
                // one member larger than the room offered: inflate aside and hand out in pieces
                Member m{scan, bsize, 12 + xlen, isize, 0};
                spill.assign(isize, 0); spill_pos = 0;
                if (!inflate_member(CD(), m, spill.data(), worker_decoder(0))) { spill.clear(); bad = true; return 0; }
                cpos = scan + bsize;
                return read_bgzf(dst, cap);
            }
            ms.push_back(Member{scan, bsize, 12 + xlen, isize, o_off});
            o_off += isize; scan += bsize;
            if (ms.size() >= 65536) break;
        }
        if (!ms.empty()) {
            const int T = (int)std::max<size_t>(1, std::min<size_t>((size_t)n_threads, ms.size() / 4));
            std::atomic<size_t> next{0}, first_bad{ms.size()};
            for (int t = 0; t < T; t++) (void)worker_decoder(t);        // created here, not in the workers
            auto work = [&](int t) {
                f2qz::Inflater *inf = worker_decoder(t);
                for (;;) {
                    const size_t i = next.fetch_add(16);
                    if (i >= ms.size()) return;
                    for (size_t j = i; j < std::min(i + 16, ms.size()); j++)
                        if (!inflate_member(CD(), ms[j], dst + ms[j].o_off, inf)) {
                            size_t cur = first_bad.load();
                            while (j < cur && !first_bad.compare_exchange_weak(cur, j)) {}
                        }
                }
            };
            if (T == 1) work(0);
            else {
                std::vector<std::thread> th;
                for (int t = 1; t < T; t++) th.emplace_back(work, t);
                work(0);
                for (auto &x : th) x.join();
            }
            const size_t fb = first_bad.load();
            if (fb < ms.size()) { bad = true; foreign = false; n = ms[fb].o_off; cpos = ms[fb].c_off; }
            else { n = o_off; cpos = scan; }
        }
        if (n == 0 && foreign && !bad) {                      // the rest is ordinary gzip: continue from that member
            std::string err;
            const uint64_t at = file_pos - (uint64_t)(CN() - scan);
            cbuf.clear(); cpos = 0;
            if (bmap) { munmap(const_cast<uint8_t *>(bmap), bmap_len); bmap = nullptr; bmap_len = 0; }
            if (open_gzip(at, err) != 0) { bad = true; return 0; }
            return zmap ? read_gzip_mm(dst, cap) : read_gzip(dst, cap);
        }
        // empty members (the BGZF end marker) produce nothing: keep going until text or the end
        if (n == 0 && !bad && !ms.empty()) return read_bgzf(dst, cap);
        return n;
    }
};
