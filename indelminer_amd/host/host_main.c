/* host_main.c -- part of the indelminer host driver (one translation unit: imhost.c includes the parts in order, so that the
 * reference-shaped helpers can stay static).  Here: usage, option parsing, main. */

/* -------------------------------------------------------------------- main -- */

static void print_help(FILE* file)
{
    /* src/indelminer.c:883-923 */
    fprintf(file, "\n");
    fprintf(file, "Program: indelminer (Call/Tag indels from a clean BAM file)\n");
    fprintf(file, "Version: %2.2f\n\n", INDELMINER_VERSION);
    fprintf(file, "Usage:\n");
    fprintf(file, "\tindelminer [options] ref.fa [indels.vcf] sample=aln.bam\n");
    fprintf(file, "where the options are\n");
    fprintf(file, "\t-h   print help and return\n");
    fprintf(file, "\n");
    fprintf(file, "\t-i, read the configuration from this file\n");
    fprintf(file, " \t-c, only analyze this chromosomal region [ALL]\n");
    fprintf(file, " \t-t, do not call indels on 3' regions of the read\n");
    fprintf(file, " \t-q, do not call indels from reads with MQ < INT [10]\n");
    fprintf(file, " \t-n, disallow indel within INT bp towards the ends [10]\n");
    fprintf(file, " \t    We ignore the 3' soft-clipping, since that is where\n");
    fprintf(file, " \t    we expect the low quality region on the reads\n");
    fprintf(file, "\t-a, in case of overlapping indels, call all of them\n");
    fprintf(file, "\t    Default is to call the indels with most support\n");
    fprintf(file, "\t-e, minimum support for an indel [2]\n");
    fprintf(file, "\t-o, output format. vcf/detailed [vcf]\n");
    fprintf(file, "\t-s, maximum size of deletion reported using split reads [1 kbp]\n");
    fprintf(file, "\t-p, maximum size of deletion reported using PE reads [1 Mbp]\n");
    fprintf(file, "\n");
    fprintf(file, "\t-k, length of the kmers to be used in alignments[6]\n");
    fprintf(file, "\t-g, number of gaps allowed in the alignments[0]\n");
    fprintf(file, "\t-f, number of differences allowed in an alignment[6]\n");
    fprintf(file, "\t-b, require at least one read with these bases on \n");
    fprintf(file, "\t    either side of the indel[30]\n");
    fprintf(file, "\n");
    fprintf(file, "Assumptions:\n");
    fprintf(file, "\tThe BAM file is coordinate sorted\n");
    fprintf(file, "\tUnless specified in a config file, insertlengths for\n");
    fprintf(file, "\treadgroups, as well as average coverage per chromosome\n");
    fprintf(file, "\tare estimated from the BAM file (which can be slow!!!),\n");
    fprintf(file, "\tas well as lead to false negatives as some of the PE\n");
    fprintf(file, "\tevidence which is accounted for in one sample,might not\n");
    fprintf(file, "\tbe accounted for in the other\n");
}

static void free_range(void* p) { free(p); }

int main(int argc, char** argv)
{
    t_is_main = 1;
    g_argv = xcalloc((size_t)argc + 1, sizeof(char*));          /* as given: the parsing below cuts the sample argument in two */
    for (int i = 0; i < argc; i++) g_argv[i] = xstrdup(argv[i]);
    {
        /* the record-at-a-time child of a pipeline run the reference aborts: its first bytes are on stdout already, and so is
         * everything it has to say on stderr until something goes wrong */
        const char* sk = getenv("INDELMINER_SKIP_STDOUT");
        const char* from = getenv("INDELMINER_HANDOFF_PARENT");       /* only the process that started this one may ask for it: a value
                                                                        * that leaked into somebody's environment drops nothing */
        if (sk && !(from && (long)getppid() == atol(from))) sk = NULL;
        if (sk) {
            g_out_skip = atoll(sk);
            unsetenv("INDELMINER_SKIP_STDOUT");
            t_out = out_cookie_open();
            if (getenv("INDELMINER_HANDOFF_QUIET") && !getenv("INDELMINER_DEBUG_HANDOFF")) {
                g_real_stderr = dup(STDERR_FILENO);
                if (!freopen("/dev/null", "w", stderr)) { }
            }
        }
    }
    O.maxdelsize = 1000; O.maxpedelsize = 1000000; O.minsupport = 2; O.klength = 6; O.numgaps = 0;
    O.outputformat = "vcf"; O.qthreshold = 10; O.ethreshold = 10; O.ethreshold_vcfcheck = 10;
    O.call_all_indels = 0; O.maxdiffsallowed = 6; O.minbalance = 30;
    const char* tie_env = getenv("INDELMINER_TIE_ORDER");       /* "expected": SURVEY.md 0.2 */
    O.tie_desc = (tie_env && strcmp(tie_env, "expected") == 0) ? 1 : 0;

    int c;
    while ((c = getopt(argc, argv, "dl:hc:e:o:k:g:x:i:s:p:tn:q:af:b:")) != -1) {
        switch (c) {
        case 'd': O.debug = 1; break;
        case 'l': break;
        case 'h': print_help(stdout); return EXIT_SUCCESS;
        case 'c': O.region = optarg; break;
        case 'e': if (sscanf(optarg, "%u", &O.minsupport) != 1) fatalf("incorrect option for -e: %s\n", optarg); break;
        case 'o': O.outputformat = optarg; break;
        case 'k': if (sscanf(optarg, "%u", &O.klength) != 1) fatalf("incorrect option for -k: %s\n", optarg); break;
        case 'f': if (sscanf(optarg, "%u", &O.maxdiffsallowed) != 1) fatalf("incorrect option for -f: %s\n", optarg); break;
        case 'g': if (sscanf(optarg, "%u", &O.numgaps) != 1) fatalf("incorrect option for -g: %s\n", optarg); break;
        case 'x': break;                                            /* accepted, unused (src/indelminer.c:793-794) */
        case 'i': O.configfile = optarg; break;
        case 's': if (sscanf(optarg, "%u", &O.maxdelsize) != 1) fatalf("incorrect option for -s: %s\n", optarg); break;
        case 'p': if (sscanf(optarg, "%u", &O.maxpedelsize) != 1) fatalf("incorrect option for -p: %s\n", optarg); break;
        case 't': break;                                            /* stored, never read (src/indelminer.c:775) */
        case 'n':
            if (sscanf(optarg, "%u", &O.ethreshold) != 1) fatalf("incorrect option for -n: %s\n", optarg);
            if (O.ethreshold < O.klength) O.ethreshold = O.klength;
            O.ethreshold_vcfcheck = O.ethreshold;
            break;
        case 'q': if (sscanf(optarg, "%d", &O.qthreshold) != 1) fatalf("incorrect option for -q: %s\n", optarg); break;
        case 'a': O.call_all_indels = 1; break;
        case 'b': if (sscanf(optarg, "%u", &O.minbalance) != 1) fatalf("incorrect option for -b: %s\n", optarg); break;
        case '?': break;
        default: print_help(stderr); return EXIT_FAILURE;
        }
    }
    forceassert(O.maxdelsize > 0);
    forceassert(O.klength > 1 && O.klength < 16);
    forceassert(strcmp(O.outputformat, "vcf") == 0 || strcmp(O.outputformat, "detailed") == 0);
    if (argc == optind) { print_help(stderr); return EXIT_FAILURE; }
    forceassert(argc - optind > 1);
    t0 = time(0);
    g_timing = getenv("INDELMINER_TIMING") != NULL;
    g_t_last = now_ms();

    const char* fasta_reference = argv[optind++];
    char* ptr = argv[optind++];
    if (strchr(ptr, '=') == NULL) {                 /* a VCF: tag its indels only (src/indelminer.c:1046-1053) */
        g_vcfname = ptr;
        O.minsupport = 1;
        O.outputformat = "vcf";
        ptr = argv[optind++];
    }
    char* samplename = ptr;
    while (*ptr != '=') ptr++;
    *ptr = 0;
    const char* bam_name = ++ptr;
    g_sample_name = samplename;
    if (g_vcfname != NULL) O.ethreshold_vcfcheck = 0;   /* src/indelminer.c:1074 */

    fprintf(stderr, "Reference fasta file: %s\n", fasta_reference);
    fprintf(stderr, "Chromosomal region  : %s\n", O.region == NULL ? "ALL" : O.region);
    fprintf(stderr, "BAM file            : %s\n", bam_name);
    if (g_vcfname != NULL) fprintf(stderr, "VCF file            : %s\n", g_vcfname);

    driver d;
    memset(&d, 0, sizeof d);
    d.depth_tid = -1;
    d.bam_name = bam_name;
    bgzf_reader* r = bgzf_open(bam_name);
    if (!r) fatalf("error in opening the file %s", bam_name);
    d.hdr = bam_header_load(r);
    if (!d.hdr) fatalf("%s is not a BAM file", bam_name);
    d.idx = bai_load(bam_name);
    if (!d.idx) fatalf("BAM indexing file is not available.");
    d.insertlengths = qhash_new(4);
    d.readpairs = qhash_new(20);

    int chromid = -1, chromstart = -1, chromstop = -1;
    if (O.region) bam_parse_region_str(d.hdr, O.region, &chromid, &chromstart, &chromstop);

    /* one process per GPU (torch.distributed.run's environment): contigs are sharded over the ranks */
    mgpu mg;
    memset(&mg, 0, sizeof mg);
    {
        const char* ws = getenv("WORLD_SIZE");
        const int world = ws ? atoi(ws) : 1;
        const char* pl = getenv("INDELMINER_PIPELINE");
        if ((world > 1 || getenv("INDELMINER_FORCE_MGPU")) && chromid == -1 && !(pl && strcmp(pl, "host") == 0)) {
            mg.world = world > 0 ? world : 1;
            mg.rank = getenv("RANK") ? atoi(getenv("RANK")) : 0;
            mg.local_rank = getenv("LOCAL_RANK") ? atoi(getenv("LOCAL_RANK")) : mg.rank;
            forceassert(mg.rank >= 0 && mg.rank < mg.world);
            g_mg = &mg; g_mg_rank = mg.rank; g_mg_local = mg.local_rank; g_mg_parts = 1;
            /* librccl prints a banner on descriptor 1: the VCF goes through part files and the saved descriptor, and
             * descriptor 1 points at stderr for the whole run */
            fflush(stdout);
            mg.out_fd = dup(1);
            if (mg.out_fd < 0 || dup2(2, 1) < 0) fatalf("cannot redirect stdout");
            mg.abort_tid = -1;
            if (g_vcfname != NULL) {
                /* annotate mode walks only the contigs the variant file names (src/indelminer.c:788) */
                mg.skip = xcalloc((size_t)d.hdr->n_targets, 1);
                for (int32_t i = 0; i < d.hdr->n_targets; i++) {
                    known_free(&g_known);
                    read_variants(g_vcfname, i, d.hdr->target_name[i], &g_known);
                    mg.skip[i] = g_known.n == 0;
                }
            }
            mg_plan(&mg, &d);                   /* before the walkers are planned: they take this rank's contigs */
        }
    }
    d.marker_floor = INT_MAX;

    /* the GPU: one context, opened by a helper thread while this thread reads the BAM (insert lengths) and the FASTA --
     * HIP start-up is 0.15-0.3 s of nothing but waiting */
    pthread_mutex_init(&d.gpu_mu, NULL); pthread_cond_init(&d.gpu_cv, NULL);
    d.gpu_pending = 1;
    if (pthread_create(&d.gpu_thread, NULL, gpu_open_thread, &d) != 0) fatalf("cannot start the GPU helper thread");
    /* the device pipeline: its walkers set their buffers up from now on.  A region run (-c) is the same pipeline over the pieces of
     * one stretch of one contig (its first piece also takes the records that reach into it from the front, as bam_fetch does;
     * mates outside the stretch and the depth around a variant are looked up in the file, like the reference does).
     * INDELMINER_PIPELINE=host is the record-at-a-time path, kept for runs the reference aborts (handoff_to_host_child). */
    walkpool_t* pool = NULL;
    {
        const char* pl0 = getenv("INDELMINER_PIPELINE");
        if (!(pl0 && strcmp(pl0, "host") == 0)) {
            if (chromid != -1) {
                g_region_tid = chromid; g_region_beg = chromstart; g_region_end = chromstop;
                /* not clipped to the contig's length: bam_fetch(tid, beg, end) with "ctg" alone means end = 2^29 and delivers records
                 * whose position lies behind the contig's end too (a whole-genome run, which fetches [0, length), never sees them) */
                if (g_region_end < g_region_beg) g_region_end = g_region_beg;
            }
            /* no config file: the insert lengths are estimated by the walk itself (run_pipeline) instead of by a pass of their own;
             * multi-GPU runs and annotate mode keep the pre-pass (the shard summaries / the serial walk need the table up front) */
            {
                const char* op = getenv("INDELMINER_ONEPASS");          /* INDELMINER_ONEPASS=0: the pre-pass of the reference's layout */
                /* several ranks: always the one walk (run_pipeline: the ranks exchange what their walks logged) */
                g_onepass = O.configfile == NULL && g_vcfname == NULL && chromid == -1 && (g_mg || !(op && strcmp(op, "0") == 0));
            }
            pool = walkpool_start(&d);
        }
    }
    if (g_mg) mg_rendezvous(&mg, &d);          /* the communicator comes up on a thread of its own, beside the FASTA read and the pre-walk */

    if (O.configfile) read_configuration(O.configfile, d.insertlengths, d.hdr);
    else if (g_onepass) { }
    else if (!g_mg) { if (chromid == -1 && pool && !getenv("INDELMINER_ESTIMATE_SERIAL")) estimate_insertlengths_threads(&d, pool->pieces, pool->n_pieces); else estimate_insertlengths(&d, chromid); }
    fprintf(stderr, "\nRead-group\tMin-value\tMax-value\n----------\t---------\t---------\n");
    for (uint32_t i = 0; i <= d.insertlengths->mask; i++)
        for (qbin* it = d.insertlengths->bins[i]; it; it = it->next)
            fprintf(stderr, "%s\t%d\t%d\n", it->name, ((int32_t*)it->val)[0], ((int32_t*)it->val)[1]);
    fprintf(stderr, "----------\t---------\t---------\n\n");
    if (!g_onepass && !(g_mg && O.configfile == NULL)) cov_print_table(d.hdr);      /* one-pass and multi-rank estimates: printed when every record has been seen */
    timestamp("Read insertlengths for the BAM file");
    phase_time("open BAM, index, insert lengths");

    const int nseq = fasta_load(fasta_reference, d.hdr->n_targets, &d.sequences, &d.seqlen, chromid);
    if (nseq < 0) fatalf("error in opening the file %s", fasta_reference);
    forceassert(nseq == d.hdr->n_targets);
    timestamp("Read the reference sequence");
    phase_time("read FASTA");

    /* the reference is in: the GPU helper (started before the insert-length pass) uploads it */
    pthread_mutex_lock(&d.gpu_mu);
    d.seq_ready = 1;
    pthread_cond_broadcast(&d.gpu_cv);
    pthread_mutex_unlock(&d.gpu_mu);


    const char* pl = getenv("INDELMINER_PIPELINE");
    const int use_pipeline = !(pl && strcmp(pl, "host") == 0);
    if (g_mg && pool->serial) {
        /* annotate mode: contigs are walked one by one behind each other's replays -- the logs come from a pass of their own */
        mg_exchange(&mg, &d, O.configfile == NULL, pool->pieces, pool->n_pieces, mg.piece_walker);
        mg_after_logs(&mg, &d);
    }
    if (use_pipeline) run_pipeline(&d, pool);
    if (g_mg) mg_finish(&mg, &d);
    /* the reference prints its header before it reads the first record (src/indelminer.c:745-754 in front of 756-): a run it
     * aborts on some record has the header on stdout.  Here the header waits for the GPU context (gpu_wait). */
    if (!use_pipeline) gpu_wait(&d);
    for (int32_t i = 0; i < d.hdr->n_targets && !use_pipeline; i++) {
        if (chromid != -1 && i != chromid) continue;
        if (g_vcfname != NULL) {
            known_free(&g_known);
            read_variants(g_vcfname, i, d.hdr->target_name[i], &g_known);
            if (g_known.n == 0) continue;           /* src/indelminer.c:788 */
        }
        if (chromid == -1) run_contig(&d, i, 0, d.hdr->target_len[i], r);
        else run_contig(&d, i, chromstart, chromstop, r);
    }

    gpu_wait(&d);
    if (t_out) fflush(t_out);
    if (g_vcfname != NULL) cpu_report();            /* annotate mode has no replay workers: its report comes here */
    if (!getenv("INDELMINER_TIDY_EXIT")) {
        /* everything is printed: the GPU context, the pinned rings and the device arrays go with the process -- tearing the
         * HIP runtime down in order costs about as long as the whole device work of a small run (leak checkers: INDELMINER_TIDY_EXIT=1) */
        fflush(stdout);
        phase_time("output flushed; the process ends (what follows is the kernel taking it apart)");
        fflush(stderr);
        _exit(EXIT_SUCCESS);
    }
    im_ctx_destroy(d.gpu);
    bgzf_close(r);
    bai_free(d.idx);
    qhash_free(d.insertlengths, free_range);
    return EXIT_SUCCESS;
}
