/*
 * ref_unit_sw.c -- TEST INFRASTRUCTURE ONLY (oracle/Makefile, target _ref/librefunits.so).
 *
 * Lets the tests call the REAL reference's static realign_with_indel (src/variant.c:1246-1424): this
 * translation unit is the reference's own variant.c compiled from where it lies (the #include below; nothing
 * is copied) plus one exported wrapper of ours.  Used by tests/golden/make_golden_units.py to make
 * function-level golden vectors for imo_sw_indel and the device support kernel.
 */
#include "variant.c"

/* reference: NUL-terminated contig; [rstart, rstop) the window check_for_indel hands over (src/variant.c:1536-1546);
 * query / qstart / qstop: the read and its aligned part; the variant: is_deletion, start, stop, alternate. */
void imref_realign_with_indel(const char* reference, int rstart, int rstop, const char* query, int qstart, int qstop,
                              int is_deletion, unsigned vstart, unsigned vstop, const char* alternate,
                              int* subs, int* indels, int* aligned)
{
    readseg seg; memset(&seg, 0, sizeof seg);
    seg.sequence = (char*)query;
    readaln rln; memset(&rln, 0, sizeof rln);
    rln.segments = &seg;
    knownvariant k; memset(&k, 0, sizeof k);
    k.type = is_deletion ? DELETION : INSERTION;
    k.start = vstart; k.stop = vstop; k.alternate = (char*)alternate;
    realign_with_indel(reference, rstart, rstop, &rln, qstart, qstop, &k, subs, indels, aligned);
}
