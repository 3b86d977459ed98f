/*
 * ref_unit_pe.c -- TEST INFRASTRUCTURE ONLY (oracle/Makefile, target _ref/librefunits.so).
 *
 * Lets the tests call the REAL reference's static process_evidence (src/indelminer.c:117-209) on a list of
 * evidence records: this translation unit is the reference's own indelminer.c, compiled from where it lies
 * (the #include below; nothing is copied), with its main renamed, plus one exported wrapper of ours that
 * builds the list the way fetch_func does (sladdhead in arrival order, src/indelminer.c:417,496-509,604) and
 * flattens the returned variants.  Used by tests/golden/make_golden_units.py to make function-level golden
 * vectors for imo_cluster_sr and the device cluster kernels.
 */
#define main imref_indelminer_main_unused
#include "indelminer.c"
#undef main

/* n evidence records in arrival order.  type[i]: 0 SPLIT_READ, 1 PAIRED_READ (PAIRED_READ needs s1/e3/maxv:
 * start of the first read, end of the second, the read group's range[1]).  out: per variant, in the order of
 * the returned list, {variantclass, evidence type, start, stop, support, member ordinals...}; used[i] = isused.
 * Returns the number of ints written, or -1 if cap is too small. */
int imref_process_evidence(int n, const int* type, const int* cls, const int* b1, const int* b2,
                           const int* s1, const int* e3, const int* maxv, int marker,
                           int* out, int cap, int* n_variants, unsigned char* used)
{
    evidence* list = NULL;
    evidence** all = ckallocz((n ? n : 1) * sizeof(evidence*));
    for (int i = 0; i < n; i++) {
        evidence* e = ckallocz(sizeof(evidence));
        e->type = type[i] ? PAIRED_READ : SPLIT_READ;
        e->variantclass = cls[i] ? DELETION : INSERTION;
        e->b1 = b1[i]; e->b2 = b2[i];
        e->mindelsize = i;                  /* carries the ordinal through (PAIRED_READ only field, not read by process_evidence) */
        if (type[i]) {
            readseg* a = ckallocz(sizeof(readseg)); a->start = s1[i]; a->end = b1[i]; a->op = BAM_CMATCH; a->oplen = (uint32_t)(b1[i] - s1[i]);
            readseg* c = ckallocz(sizeof(readseg)); c->start = b2[i]; c->end = e3[i]; c->op = BAM_CMATCH; c->oplen = (uint32_t)(e3[i] - b2[i]);
            e->aln1 = a; e->aln3 = c; e->max = maxv[i];
        }
        all[i] = e;
        sladdhead(&list, e);
    }
    variant* vs = process_evidence(&list, 0, marker);
    int w = 0, nv = 0;
    for (variant* v = vs; v; v = v->next) {
        if (w + 5 + (int)v->support > cap) return -1;
        out[w++] = (int)v->type; out[w++] = (int)v->evdnctype; out[w++] = (int)v->start; out[w++] = (int)v->stop; out[w++] = (int)v->support;
        for (uint i = 0; i < v->support; i++) out[w++] = v->evidence[i]->mindelsize;
        nv++;
    }
    for (int i = 0; i < n; i++) used[i] = all[i]->isused ? 1 : 0;
    *n_variants = nv;
    return w;
}
