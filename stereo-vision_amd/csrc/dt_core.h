// Delaunay divide and conquer, the part that is pure pointer work: triangle records, leaves and the merge of two
// triangulated halves with Triangle's tie rules and record order (libelas/src/triangle.cpp:5638-5934 mergehulls,
// 5953-6103 divconqrecurse).  Compiled twice from this one header:
//   * by hipcc into k_delaunay (csrc/elas_stage_kernels.hip): one lane per merge, records in LDS or in L2;
//   * by g++ into tests/cxx/dt_core_check.cpp: the same functions over plain memory, run bottom-up by depth like the
//     kernel does and compared with csrc/delaunay.cpp (itself checked against the real Triangle), so that the
//     read-ordering arguments below are exercised on the CPU on degenerate point sets before any GPU sees them.
//
// Round 5: WHOLE-RECORD reads.  The walk along a seam is a chain of dependent reads, and on one lane the round trip
// of a read (LDS ~130 cycles, L2 ~700) is what a step costs.  Round 2-4's form fetched field by field (neighbour
// handle -> corner index -> coordinates of that vertex: three dependent trips for "the apex of the triangle behind
// this edge").  Here a record carries the coordinates of its corners inline and is fetched WHOLE -- three corners
// (index + packed x|y) and three neighbour handles in one trip -- and the merge is arranged so that
//   * read-only walks (the bottommost/topmost rotation of horizontal cuts, the lower common tangent, the rotation
//     back) advance all their hulls in ONE loop, one trip per iteration instead of one per hull and field;
//   * a seam step costs three trips (new candidate handle, its record, the record behind it) instead of six,
//     an edge flip one instead of four.
// Every read that the sequential algorithm performs AFTER a write is still performed after that write (same lane,
// program order), with one exception that is guarded explicitly: inside a flip the record behind the next candidate
// edge is requested before the flip's last stores are issued, and requested again when it is one of the three
// records those stores go to.  Nothing is assumed about which records can coincide (tiny and collinear hulls share
// several edges between the same two ghost records).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define DT_FN __device__ __forceinline__
#else
#define DT_FN inline
#endif

namespace svh {
namespace dt {

struct Vtx {
    int id;   // input index, -1 = the ghost vertex
    int xy;   // x | y << 16  (0 for the ghost)
};
DT_FN int vx(const Vtx& v) { return v.xy & 0xffff; }
DT_FN int vy(const Vtx& v) { return (int)((unsigned)v.xy >> 16); }
DT_FN Vtx ghost() { Vtx g; g.id = -1; g.xy = 0; return g; }

// one triangle record in registers; slot k = corner k / the neighbour across the edge of orientation k
struct Rec {
    int id0, id1, id2;
    int xy0, xy1, xy2;
    unsigned n0, n1, n2;
};

// handle = record * 4 + orientation (0..2); 0 = "outer space" (record 0 is a sink nobody's result depends on)
DT_FN int p1(int o) { return (0x09 >> (2 * o)) & 3; }   // (o + 1) % 3
DT_FN int p2(int o) { return (0x12 >> (2 * o)) & 3; }   // (o + 2) % 3
DT_FN unsigned hnext(unsigned h) { return (h & ~3u) | (unsigned)p1(h & 3); }
DT_FN unsigned hprev(unsigned h) { return (h & ~3u) | (unsigned)p2(h & 3); }
DT_FN bool same_rec(unsigned a, unsigned b) { return ((a ^ b) >> 2) == 0; }

DT_FN int sel3(int o, int a, int b, int c) { return o == 0 ? a : (o == 1 ? b : c); }
DT_FN Vtx rcorner(const Rec& r, int o) {
    Vtx v;
    v.id = sel3(o, r.id0, r.id1, r.id2);
    v.xy = sel3(o, r.xy0, r.xy1, r.xy2);
    return v;
}
DT_FN unsigned rnbr(const Rec& r, int o) { return (unsigned)sel3(o, (int)r.n0, (int)r.n1, (int)r.n2); }
// of the triangle whose record is r, seen through handle h
DT_FN Vtx r_apex(const Rec& r, unsigned h) { return rcorner(r, h & 3); }
DT_FN Vtx r_org(const Rec& r, unsigned h) { return rcorner(r, p1(h & 3)); }
DT_FN Vtx r_dest(const Rec& r, unsigned h) { return rcorner(r, p2(h & 3)); }
DT_FN unsigned r_sym(const Rec& r, unsigned h) { return rnbr(r, h & 3); }

// ---- storages.  Interface: load(h) whole record, sym(h) one neighbour field, set_corner, bond, make_rec, ids_of.
//   MeshG  global memory (L2-resident), 32-bit fields, three 16-byte rows per record (ids, xys, nbr);
//   MeshL  LDS, 24 bytes per record: xy0 xy1 xy2 (32 bit each) | id0 id1 id2 | n0 n1 n2 (16 bit each);
//          used when 48 bytes per point fit the block's allocation (ids and handles below 2^15 / 2^16).
struct alignas(16) Row4 {
    int x, y, z, w;
};
struct MeshG {
    int* ids;
    int* xys;
    unsigned* nbr;
    DT_FN Rec load(unsigned h) const {
        const unsigned t4 = h & ~3u;
        const Row4 a = *reinterpret_cast<const Row4*>(ids + t4);
        const Row4 b = *reinterpret_cast<const Row4*>(xys + t4);
        const Row4 c = *reinterpret_cast<const Row4*>(nbr + t4);
        Rec r;
        r.id0 = a.x; r.id1 = a.y; r.id2 = a.z;
        r.xy0 = b.x; r.xy1 = b.y; r.xy2 = b.z;
        r.n0 = (unsigned)c.x; r.n1 = (unsigned)c.y; r.n2 = (unsigned)c.z;
        return r;
    }
    DT_FN unsigned sym(unsigned h) const { return nbr[h]; }
    DT_FN void set_corner(unsigned t4, int k, const Vtx& v) const {
        ids[t4 + k] = v.id;
        xys[t4 + k] = v.xy;
    }
    DT_FN void bond(unsigned a, unsigned b) const {
        nbr[a] = b;
        nbr[b] = a;
    }
    DT_FN unsigned make_rec(int t) const {
        Row4 m1; m1.x = m1.y = m1.z = -1; m1.w = 0;
        Row4 z; z.x = z.y = z.z = z.w = 0;
        *reinterpret_cast<Row4*>(ids + 4 * (size_t)t) = m1;
        *reinterpret_cast<Row4*>(xys + 4 * (size_t)t) = z;
        *reinterpret_cast<Row4*>(nbr + 4 * (size_t)t) = z;
        return (unsigned)t * 4u;
    }
};
struct MeshL {
    unsigned char* base;   // 8-byte aligned
    static constexpr unsigned kBytes = 24;
    DT_FN Rec load(unsigned h) const {
        const uint64_t* p = reinterpret_cast<const uint64_t*>(base + (h >> 2) * kBytes);
        const uint64_t w0 = p[0], w1 = p[1], w2 = p[2];
        Rec r;
        r.xy0 = (int)(uint32_t)w0;
        r.xy1 = (int)(uint32_t)(w0 >> 32);
        r.xy2 = (int)(uint32_t)w1;
        r.id0 = (int)(int16_t)(uint16_t)(w1 >> 32);     // 0xffff -> -1
        r.id1 = (int)(int16_t)(uint16_t)(w1 >> 48);
        r.id2 = (int)(int16_t)(uint16_t)w2;
        r.n0 = (unsigned)(uint16_t)(w2 >> 16);
        r.n1 = (unsigned)(uint16_t)(w2 >> 32);
        r.n2 = (unsigned)(uint16_t)(w2 >> 48);
        return r;
    }
    DT_FN unsigned sym(unsigned h) const {
        return *reinterpret_cast<const uint16_t*>(base + (h >> 2) * kBytes + 18 + 2 * (h & 3));
    }
    DT_FN void set_corner(unsigned t4, int k, const Vtx& v) const {
        unsigned char* q = base + (t4 >> 2) * kBytes;
        *reinterpret_cast<uint32_t*>(q + 4 * k) = (uint32_t)v.xy;
        *reinterpret_cast<uint16_t*>(q + 12 + 2 * k) = (uint16_t)v.id;
    }
    DT_FN void bond(unsigned a, unsigned b) const {
        *reinterpret_cast<uint16_t*>(base + (a >> 2) * kBytes + 18 + 2 * (a & 3)) = (uint16_t)b;
        *reinterpret_cast<uint16_t*>(base + (b >> 2) * kBytes + 18 + 2 * (b & 3)) = (uint16_t)a;
    }
    DT_FN unsigned make_rec(int t) const {
        uint64_t* p = reinterpret_cast<uint64_t*>(base + (unsigned)t * kBytes);
        p[0] = 0;
        p[1] = 0xffffffff00000000ull;   // id0 = id1 = ghost
        p[2] = 0x000000000000ffffull;   // id2 = ghost, no neighbours
        return (unsigned)t * 4u;
    }
};

template <class M> DT_FN void set_org(const M& m, unsigned h, const Vtx& v) { m.set_corner(h & ~3u, p1(h & 3), v); }
template <class M> DT_FN void set_dest(const M& m, unsigned h, const Vtx& v) { m.set_corner(h & ~3u, p2(h & 3), v); }
template <class M> DT_FN void set_apex(const M& m, unsigned h, const Vtx& v) { m.set_corner(h & ~3u, h & 3, v); }

// exact predicates: coordinates are integers in [0, 2^14)
DT_FN int ccw(const Vtx& a, const Vtx& b, const Vtx& c) {
    const int l = (vx(a) - vx(c)) * (vy(b) - vy(c));
    const int r = (vy(a) - vy(c)) * (vx(b) - vx(c));
    return l > r ? 1 : (l < r ? -1 : 0);
}
// a * b + c on 32-bit signed factors: one v_mad_i64_i32 (hipcc splits the product of a sign-extended and a
// zero-extended 32-bit value into two v_mad_u64_u32 and three moves)
DT_FN long long mad64(int a, int b, long long c) {
#if defined(__HIP_DEVICE_COMPILE__)
    long long r;
    unsigned long long carry;
    asm("v_mad_i64_i32 %0, %1, %2, %3, %4" : "=v"(r), "=s"(carry) : "v"(a), "v"(b), "v"(c));
    return r;
#else
    return (long long)a * b + c;
#endif
}
// kPlain: the multiply-adds in plain C++ (a wave that runs ONE merge on all its lanes keeps the walk in scalar
// registers, and inline assembly with vector operands would make everything behind the test lane-dependent)
template <bool kPlain = false>
DT_FN int incircle(const Vtx& a, const Vtx& b, const Vtx& c, const Vtx& d) {
    if (kPlain) {
        const int adx = vx(a) - vx(d), ady = vy(a) - vy(d);
        const int bdx = vx(b) - vx(d), bdy = vy(b) - vy(d);
        const int cdx = vx(c) - vx(d), cdy = vy(c) - vy(d);
        const int al = adx * adx + ady * ady, bl = bdx * bdx + bdy * bdy, cl = cdx * cdx + cdy * cdy;
        const long long det = (long long)al * (bdx * cdy - cdx * bdy) + (long long)bl * (cdx * ady - adx * cdy) +
                              (long long)cl * (adx * bdy - bdx * ady);
        return det > 0 ? 1 : (det < 0 ? -1 : 0);
    }
    const int adx = vx(a) - vx(d), ady = vy(a) - vy(d);
    const int bdx = vx(b) - vx(d), bdy = vy(b) - vy(d);
    const int cdx = vx(c) - vx(d), cdy = vy(c) - vy(d);
    const int al = adx * adx + ady * ady, bl = bdx * bdx + bdy * bdy, cl = cdx * cdx + cdy * cdy;   // < 2^29
    const long long det = mad64(al, bdx * cdy - cdx * bdy, mad64(bl, cdx * ady - adx * cdy, mad64(cl, adx * bdy - bdx * ady, 0)));
    return det > 0 ? 1 : (det < 0 ? -1 : 0);
}

// leaves of the recursion (triangle.cpp:5953-6103, n == 2 and n == 3); records ctr .. ctr + (n == 2 ? 1 : 3)
template <class M>
DT_FN void dt_leaf(const M& m, const int* order, const int* oxy, int s, int n, int ctr, unsigned* farleft,
                   unsigned* farright) {
    Vtx a[3];
    for (int k = 0; k < 3; k++) {
        a[k].id = k < n ? order[s + k] : -1;
        a[k].xy = k < n ? oxy[s + k] : 0;
    }
    if (n == 2) {
        unsigned L = m.make_rec(ctr), R = m.make_rec(ctr + 1);
        set_org(m, L, a[0]);
        set_dest(m, L, a[1]);
        set_org(m, R, a[1]);
        set_dest(m, R, a[0]);
        m.bond(L, R);
        L = hprev(L); R = hnext(R);
        m.bond(L, R);
        L = hprev(L); R = hnext(R);
        m.bond(L, R);
        *farright = R;
        *farleft = hprev(R);
        return;
    }
    unsigned mid = m.make_rec(ctr), t1 = m.make_rec(ctr + 1), t2 = m.make_rec(ctr + 2), t3 = m.make_rec(ctr + 3);
    const int area = ccw(a[0], a[1], a[2]);
    if (area == 0) {
        set_org(m, mid, a[0]); set_dest(m, mid, a[1]);
        set_org(m, t1, a[1]);  set_dest(m, t1, a[0]);
        set_org(m, t2, a[2]);  set_dest(m, t2, a[1]);
        set_org(m, t3, a[1]);  set_dest(m, t3, a[2]);
        m.bond(mid, t1);
        m.bond(t2, t3);
        mid = hnext(mid); t1 = hprev(t1); t2 = hnext(t2); t3 = hprev(t3);
        m.bond(mid, t3);
        m.bond(t1, t2);
        mid = hnext(mid); t1 = hprev(t1); t2 = hnext(t2); t3 = hprev(t3);
        m.bond(mid, t1);
        m.bond(t2, t3);
        *farleft = t1;
        *farright = t2;
    } else {
        const Vtx p = area > 0 ? a[1] : a[2];
        const Vtx q = area > 0 ? a[2] : a[1];
        set_org(m, mid, a[0]); set_dest(m, t1, a[0]); set_org(m, t3, a[0]);
        set_dest(m, mid, p);   set_org(m, t1, p);     set_dest(m, t2, p);
        set_apex(m, mid, q);   set_org(m, t2, q);     set_dest(m, t3, q);
        m.bond(mid, t1);
        mid = hnext(mid);
        m.bond(mid, t2);
        mid = hnext(mid);
        m.bond(mid, t3);
        t1 = hprev(t1); t2 = hnext(t2);
        m.bond(t1, t2);
        t1 = hprev(t1); t3 = hprev(t3);
        m.bond(t1, t3);
        t2 = hnext(t2); t3 = hprev(t3);
        m.bond(t2, t3);
        *farleft = t1;
        *farright = area > 0 ? t2 : hnext(t1);
    }
}

#ifndef DT_CHECK_SHORTCUT
#define DT_CHECK_SHORTCUT(cond) ((void)0)
#endif

// Merge of two triangulated halves (triangle.cpp:5638-5934); the two seam records are ctr and ctr + 1.
// kShort: take the next candidate handle of a seam step from the candidate's record as it was read when the candidate
// was chosen, when no flip has rewritten that record since (saves the step's first trip).  The record ranges of the
// two halves are disjoint, a seam edge of the other side writes into the record of an EARLIER candidate only, and a
// record is never its own neighbour; the CPU check compares the shortcut with the fresh read on every step.
template <bool kShort, class M, bool kPlain = false>
DT_FN void dt_merge(const M& m, unsigned* farleft_io, unsigned innerleft, unsigned innerright, unsigned* farright_io,
                    int axis, int ctr) {
    unsigned farleft = *farleft_io, farright = *farright_io;
    Rec RIL = m.load(innerleft), RIR = m.load(innerright);
    Rec RFL = m.load(farleft), RFR = m.load(farright);
    Vtx il_dest = r_dest(RIL, innerleft), il_apex = r_apex(RIL, innerleft);
    Vtx ir_org = r_org(RIR, innerright), ir_apex = r_apex(RIR, innerright);
    if (axis == 1) {
        // horizontal cut: the extreme handles go from leftmost / rightmost to bottommost / topmost.  Four walks
        // over records nobody writes: all four advance in one loop, one trip per iteration.
        Vtx fl_pt = r_org(RFL, farleft), fl_apex = r_apex(RFL, farleft);
        Vtx fr_pt = r_dest(RFR, farright);
        unsigned chkB = r_sym(RIL, innerleft), chkD = r_sym(RFR, farright);
        Rec RB = m.load(chkB), RD = m.load(chkD);
        Vtx cvB = r_apex(RB, chkB), cvD = r_apex(RD, chkD);
        for (;;) {
            const bool gA = vy(fl_apex) < vy(fl_pt);
            const bool gB = vy(cvB) > vy(il_dest);
            const bool gC = vy(ir_apex) < vy(ir_org);
            const bool gD = vy(cvD) > vy(fr_pt);
            if (!(gA | gB | gC | gD)) break;
            // where each walk goes (a walk that stands still asks for the record it already holds)
            const unsigned hA = gA ? rnbr(RFL, p1(farleft & 3)) : farleft;          // sym(hnext(farleft))
            const unsigned nilB = gB ? hnext(chkB) : innerleft;                       // innerleft moves into chk's record
            const unsigned hB = gB ? rnbr(RB, nilB & 3) : chkB;                       // sym(new innerleft)
            const unsigned hC = gC ? rnbr(RIR, p1(innerright & 3)) : innerright;      // sym(hnext(innerright))
            const unsigned nfrD = gD ? hnext(chkD) : farright;
            const unsigned hD = gD ? rnbr(RD, nfrD & 3) : chkD;
            const Rec nA = m.load(hA), nB = m.load(hB), nC = m.load(hC), nD = m.load(hD);
            if (gA) { farleft = hA; fl_pt = fl_apex; RFL = nA; fl_apex = r_apex(RFL, farleft); }
            if (gB) { innerleft = nilB; RIL = RB; il_apex = il_dest; il_dest = cvB; chkB = hB; RB = nB; cvB = r_apex(RB, chkB); }
            if (gC) { innerright = hC; ir_org = ir_apex; RIR = nC; ir_apex = r_apex(RIR, innerright); }
            if (gD) { farright = nfrD; RFR = RD; fr_pt = cvD; chkD = hD; RD = nD; cvD = r_apex(RD, chkD); }
        }
    }
    // lower common tangent: both hulls step in one iteration (the right test sees the left move, as in the
    // sequential form), one trip per iteration
    for (;;) {
        const bool gL = ccw(il_dest, il_apex, ir_org) > 0;
        const Vtx nd = gL ? il_apex : il_dest;
        const bool gR = ccw(ir_apex, ir_org, nd) > 0;
        if (!(gL | gR)) break;
        const unsigned hL = gL ? rnbr(RIL, p2(innerleft & 3)) : innerleft;        // sym(hprev(innerleft))
        const unsigned hR = gR ? rnbr(RIR, p1(innerright & 3)) : innerright;      // sym(hnext(innerright))
        const Rec nL = m.load(hL), nR = m.load(hR);
        if (gL) { innerleft = hL; il_dest = il_apex; RIL = nL; il_apex = r_apex(RIL, innerleft); }
        if (gR) { innerright = hR; ir_org = ir_apex; RIR = nR; ir_apex = r_apex(RIR, innerright); }
    }

    unsigned lcand = r_sym(RIL, innerleft), rcand = r_sym(RIR, innerright);   // (nothing has been written yet)
    unsigned base = m.make_rec(ctr);
    m.bond(base, innerleft);
    base = hnext(base);
    m.bond(base, innerright);
    base = hnext(base);
    set_org(m, base, ir_org);
    set_dest(m, base, il_dest);
    if (il_dest.id == r_org(RFL, farleft).id) farleft = hnext(base);      // (corners of a hull record: unwritten so far)
    if (ir_org.id == r_dest(RFR, farright).id) farright = hprev(base);

    Vtx lowerleft = il_dest, lowerright = ir_org;
    Rec RL = m.load(lcand), RR = m.load(rcand);       // after the stores above, in program order
    bool rl_ok = true, rr_ok = true;                  // RL / RR still equal the memory (no flip on that side since)
    Vtx upperleft = r_apex(RL, lcand), upperright = r_apex(RR, rcand);
    // The triangle behind each candidate edge (nx*, its apex nap*) is kept while that side stands still, as in
    // rounds 2-4: the two hulls are disjoint records.  Its RECORD (the flip's topc / sidec) is read again at the
    // top of every step, beside the moving side's read: a seam edge can write a neighbour field of it when two
    // edges of the candidate triangle lie against the same record (tiny hulls).
    unsigned nxL = 0, nxR = 0;
    Vtx napL = ghost(), napR = ghost();
    bool haveL = false, haveR = false;
    for (;;) {
        if (!haveL) nxL = rnbr(RL, p2(lcand & 3));    // sym(hprev(lcand)); RL is as fresh as this read would be
        if (!haveR) nxR = rnbr(RR, p1(rcand & 3));    // sym(hnext(rcand))
        Rec NL = m.load(nxL), NR = m.load(nxR);
        if (!haveL) napL = r_apex(NL, nxL);
        if (!haveR) napR = r_apex(NR, nxR);
        haveL = haveR = true;
        const bool leftdone = ccw(upperleft, lowerleft, lowerright) <= 0;
        const bool rightdone = ccw(upperright, lowerleft, lowerright) <= 0;
        if (leftdone && rightdone) break;
        if (!leftdone && napL.id >= 0) {
            // strip left-side edges that fail the in-circle test (flips in place)
            bool bad = incircle<kPlain>(lowerleft, lowerright, upperleft, napL) > 0;
            while (bad) {
                rl_ok = false;
                unsigned nx = hnext(nxL);
                const unsigned topc = r_sym(NL, nx);
                nx = hnext(nx);
                const unsigned sidec = r_sym(NL, nx);
                m.bond(nx, topc);
                m.bond(lcand, sidec);
                lcand = hnext(lcand);
                const unsigned outerc = m.sym(lcand);   // (after the bonds: tiny hulls share several edges)
                Rec NS = m.load(sidec);                 // requested before the stores below ...
                nx = hprev(nx);
                m.bond(nx, outerc);
                set_org(m, lcand, lowerleft);
                set_dest(m, lcand, ghost());
                set_apex(m, lcand, napL);
                set_org(m, nx, ghost());
                set_dest(m, nx, upperleft);
                set_apex(m, nx, napL);
                // ... which go to the records of outerc, lcand and nx: again if it is one of them
                if (same_rec(sidec, outerc) || same_rec(sidec, lcand) || same_rec(sidec, nx)) NS = m.load(sidec);
                upperleft = napL;
                nxL = sidec;
                NL = NS;
                napL = r_apex(NL, nxL);
                bad = napL.id >= 0 && incircle<kPlain>(lowerleft, lowerright, upperleft, napL) > 0;
            }
        }
        if (!rightdone && napR.id >= 0) {
            bool bad = incircle<kPlain>(lowerleft, lowerright, upperright, napR) > 0;
            while (bad) {
                rr_ok = false;
                unsigned nx = hprev(nxR);
                const unsigned topc = r_sym(NR, nx);
                nx = hprev(nx);
                const unsigned sidec = r_sym(NR, nx);
                m.bond(nx, topc);
                m.bond(rcand, sidec);
                rcand = hprev(rcand);
                const unsigned outerc = m.sym(rcand);
                Rec NS = m.load(sidec);
                nx = hnext(nx);
                m.bond(nx, outerc);
                set_org(m, rcand, ghost());
                set_dest(m, rcand, lowerright);
                set_apex(m, rcand, napR);
                set_org(m, nx, upperright);
                set_dest(m, nx, ghost());
                set_apex(m, nx, napR);
                if (same_rec(sidec, outerc) || same_rec(sidec, rcand) || same_rec(sidec, nx)) NS = m.load(sidec);
                upperright = napR;
                nxR = sidec;
                NR = NS;
                napR = r_apex(NR, nxR);
                bad = napR.id >= 0 && incircle<kPlain>(lowerleft, lowerright, upperright, napR) > 0;
            }
        }
        if (leftdone || (!rightdone && incircle<kPlain>(upperleft, lowerleft, lowerright, upperright) > 0)) {
            // new edge lowerleft -> upperright
            m.bond(base, rcand);
            const unsigned nb = hprev(rcand);
            set_dest(m, nb, lowerleft);
            lowerright = upperright;
            unsigned nc;
            if (kShort && rr_ok && !same_rec(base, rcand)) {
                nc = rnbr(RR, nb & 3);
                DT_CHECK_SHORTCUT(nc == m.sym(nb));
            } else {
                nc = m.sym(nb);
            }
            base = nb;
            rcand = nc;
            RR = m.load(rcand);
            rr_ok = true;
            upperright = r_apex(RR, rcand);
            haveR = false;
        } else {
            // new edge upperleft -> lowerright (also on a co-circular tie)
            m.bond(base, lcand);
            const unsigned nb = hnext(lcand);
            set_org(m, nb, lowerright);
            lowerleft = upperleft;
            unsigned nc;
            if (kShort && rl_ok && !same_rec(base, lcand)) {
                nc = rnbr(RL, nb & 3);
                DT_CHECK_SHORTCUT(nc == m.sym(nb));
            } else {
                nc = m.sym(nb);
            }
            base = nb;
            lcand = nc;
            RL = m.load(lcand);
            rl_ok = true;
            upperleft = r_apex(RL, lcand);
            haveL = false;
        }
    }
    // both hulls are done: the top bounding triangle
    unsigned top = m.make_rec(ctr + 1);
    set_org(m, top, lowerleft);
    set_dest(m, top, lowerright);
    m.bond(top, base);
    top = hnext(top);
    m.bond(top, rcand);
    top = hnext(top);
    m.bond(top, lcand);
    if (axis == 1) {
        // back to leftmost / rightmost anchors: two walks, nothing is written any more
        Rec RF = m.load(farleft), RG = m.load(farright);
        Vtx fl_pt = r_org(RF, farleft);
        Vtx fr_pt = r_dest(RG, farright), fr_apex = r_apex(RG, farright);
        unsigned chk = r_sym(RF, farleft);
        Rec RC = m.load(chk);
        Vtx cv = r_apex(RC, chk);
        for (;;) {
            const bool gE = vx(cv) < vx(fl_pt);
            const bool gF = vx(fr_apex) > vx(fr_pt);
            if (!(gE | gF)) break;
            const unsigned nfl = gE ? hprev(chk) : farleft;
            const unsigned hE = gE ? rnbr(RC, nfl & 3) : chk;                        // sym(new farleft)
            const unsigned hF = gF ? rnbr(RG, p2(farright & 3)) : farright;          // sym(hprev(farright))
            const Rec nE = m.load(hE), nF = m.load(hF);
            if (gE) { farleft = nfl; RF = RC; fl_pt = cv; chk = hE; RC = nE; cv = r_apex(RC, chk); }
            if (gF) { farright = hF; fr_pt = fr_apex; RG = nF; fr_apex = r_apex(RG, farright); }
        }
    }
    *farleft_io = farleft;
    *farright_io = farright;
}

// node (s, n) reached from the root (0, m) along the top `depth` bits of `path` (MSB first);
// returns false when a leaf is met before `depth`.  base = first record of the node.
DT_FN bool dt_descend(int m, int depth, unsigned path, int* s, int* n, int* base) {
    int ss = 0, nn = m, bb = 1;
    for (int k = depth - 1; k >= 0; k--) {
        if (nn <= 3) return false;
        const int h = nn >> 1;
        if ((path >> k) & 1) {
            bb += 2 * h - 2;
            ss += h;
            nn -= h;
        } else {
            nn = h;
        }
    }
    *s = ss; *n = nn; *base = bb;
    return true;
}
// segment of position i at `depth` (stops at leaves): start, size
DT_FN void dt_segment(int m, int depth, int i, int* s, int* n) {
    int ss = 0, nn = m;
    for (int k = 0; k < depth && nn > 3; k++) {
        const int h = nn >> 1;
        if (i < ss + h) nn = h;
        else { ss += h; nn -= h; }
    }
    *s = ss; *n = nn;
}

// one node of the bottom-up recursion: a leaf, or the merge of its two children (hull handles of the children's
// depth in cfl / cfr, by first vertex)
template <bool kShort, class M, bool kPlain = false>
DT_FN void dt_node(const M& mesh, int m, int d, unsigned j, const int* order, const int* oxy, const unsigned* cfl,
                   const unsigned* cfr, unsigned* fl, unsigned* fr) {
    int s, n, base;
    if (!dt_descend(m, d, j, &s, &n, &base)) return;
    unsigned a, b;
    if (n <= 3) {
        dt_leaf(mesh, order, oxy, s, n, base, &a, &b);
    } else {
        const int h = n >> 1;
        a = cfl[s];
        b = cfr[s + h];
        dt_merge<kShort, M, kPlain>(mesh, &a, cfr[s], cfl[s + h], &b, d & 1, base + 2 * n - 4);
    }
    fl[s] = a;
    fr[s] = b;
}

}   // namespace dt
}   // namespace svh
