// cli_main.cpp -- `panSVR aln` / `panSVR fc_aln` on the MI355X engine: the host side of the reference's
// three-stage pipeline (load_reads -> [engine] -> output_results; src/PanSVgenerateVCF/read_realignment.cpp:26-176)
// above the C ABI of include/psvr_engine.h.  Same options, positional arguments, stderr progress
// lines and SAM/BAM records as the reference; every other sub-command of panSVR is out of scope.
//
// It links libpsvr_engine.so only through psvr_engine.h.  Host-side pieces: fastq_batch.h (step 0: batches parsed straight
// into page-locked upload buffers), sam_emit.h + bam_writer.h (step 2: records of the pairs that are written).
//
// Multi-GPU (`--devices 0,1,...`): the index is resident on every device (one host upload, then device-to-device copies),
// every batch is cut into contiguous blocks -- pair i of n goes to device floor(i * D / n), kt_for's contract of independent
// items (clib/kthread.c:43-86) with the order kept -- and because the reference draws from ONE rand()/random_r sequence in input
// order, block d is moved to start where block d-1 ended (psvr_engine_rebase) before the ordered gather into step 2.
#define PSVR_BGZF_ON_DEVICE 1
#include <getopt.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <sys/time.h>
#include <time.h>
#include <unistd.h>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include "../../include/psvr_engine.h"
#include "host_io.h"
#include "fastq_batch.h"
#include "sam_emit.h"
#include "bam_writer.h"
#include "index_build.h"
#include "signal_step.h"
#include "bam_sort.h"

using namespace psvr;

struct Opt {
	int thread_n = 4;
	int gap_open = 16, gap_ex = 1, gap_open2 = 32, gap_ex2 = 0, match = 2, mismatch = 12, zdrop = 400, bw = 500;
	std::string out = "./output.bam", out_ori = "./output_ori.bam";
	bool not_ori = false, sam = false;
	long long max_use_read = 0x7fffffff;
	std::string index_dir, reads, header;
	std::string records;      // --records FILE: one JSON line per pair (what the parity tests compare)
	bool trace = false;
	std::vector<int> devices = {0};
	long long batch_pairs = 2000000;       // N_NEEDED, rr.cpp:24
	long long batch_bases = 100000000;     // MAX_read_size, rr.cpp:109 (333 334 pairs of 150 bp: the limit that actually binds)
	long long sub_pairs = 65536;           // a batch travels through the four stages in pieces of this many pairs (0 = whole batches): three
	                                       // reference-sized batches do not fill a four-stage pipeline, forty pieces do
	bool sig_all = false, sig_discard = false;   // BAM input: fc_signal's -D / -U
	int bam_level = -1;                          // zlib level of the BGZF blocks (-1 = zlib's default, what htslib's "wb" uses)
	bool bgzf_device = false;                    // the main file's BGZF blocks compressed on the first device (psvr_bgzf_compress)
};

static int usage()
{
	fprintf(stderr,
	        "\n  Usage:     panSVR  aln|fc_aln  [Options] <IndexDir> [ReadFiles.fa][ori_header_fn.sam]>\n"
	        "  Basic:   \n"
	        "    <IndexDir>      FOLDER   the directory contains index (or the anchor FASTA: the index is then built in GPU memory)\n"
	        "    [ReadFiles.fa]  FILES    reads files, FASTQ(A) format (or fq.gz), read 1 and 2 of a pair stored together ('-' = stdin),\n"
	        "                             or a *.bam: the signal step then runs in-process ([ori_header.sam] is written;\n"
	        "                             -N / -D / -U as in fc_signal: name-sorted input / all pairs are signals / drop fully matching pairs)\n"
	        "                             Using [signal] command to generate this type of file\n"
	        "    [ori_header.sam]  FILES  Header file of original BAM/CRAM file\n"
	        "  Options:\n"
	        "    -t, --thread            INT  host threads for parsing / formatting / compression (the alignment runs on the GPU) [4]\n"
	        "    -O, --gap-open1         INT  Gap open penalty 1 [16]\n"
	        "    -P, --gap-open2         INT  Gap open penalty 2 [32]\n"
	        "    -E, --gap-extension1    INT  Gap extension penalty 1 [1]\n"
	        "    -F, --gap-extension2    INT  Gap extension penalty 2 [0]\n"
	        "    -M, --match-score       INT  Match score [2]\n"
	        "    -m, --mis-score         INT  Mismatch score [12]\n"
	        "    -z, --zdrop             INT  Z-drop score [400]\n"
	        "    -w, --band-width        INT  parsed and ignored like the reference (DP band is fixed at 200) [500]\n"
	        "    -o, --output            STR  Output file [./output.bam]\n"
	        "    -p, --output_signal_ori STR  Reads not fully aligned by aligner nor re-aligner [./output_ori.bam]\n"
	        "    -Q, --not-ori                NOT output original result when score of ORI is bigger\n"
	        "    -S, --SAM                    Output as SAM, default is BAM\n"
	        "    -R, --max_use_read      INT  Max number of read pairs to align\n"
	        "        --devices           LIST HIP devices, e.g. 0,1,2,3 or 0-7: every batch is split over them in input order [0]\n"
	        "        --device            INT  the same for one device\n"
	        "        --batch             INT  read pairs per batch [2000000]\n"
	        "        --batch-bases       INT  bases per batch (the reference stops a batch at 100 MB of bases) [100000000]\n"
	        "        --sub-batch         INT  pairs per pipeline piece of a batch, 0 = whole batches (results do not depend on it) [65536]\n"
	        "        --compress-level    INT  zlib level of the BAM output's BGZF blocks, 0-9 (1 is ~3x faster than the default) [-1 = default, like htslib]\n"
	        "        --bgzf-fast              BGZF blocks from the built-in encoder on the -t threads (1.6x the speed of level 1, blocks ~10%% larger)\n"
	        "        --bgzf-device            compress the BAM output's BGZF blocks on the GPU (a lane per block; ~10 %% larger than zlib level 1,\n"
	        "                                 the host's deflate is what bounds the BAM route otherwise)\n"
	        "        --records           STR  dump per-pair decision records (JSON lines) for parity checks\n"
	        "        --trace                  add per-strand seed/chain hashes to --records\n\n");
	return 1;
}

static double cputime() { return (double)clock() / CLOCKS_PER_SEC; }
static double walltime()
{
	struct timeval tv;
	gettimeofday(&tv, nullptr);
	return tv.tv_sec + 1e-6 * tv.tv_usec;
}

// one output file: SAM text (-S) or BAM (default, like the reference's init_run)
struct OutFile {
	FILE *sam = nullptr;
	psvr::BamWriter bam;
	bool is_bam = false;
	bool open(const std::string &fn, bool as_bam, const HeaderInfo &H, int threads, int level = -1)
	{
		is_bam = as_bam;
		if (!as_bam) { sam = fopen(fn.c_str(), "w"); if (sam) { setvbuf(sam, nullptr, _IOFBF, 1 << 22); fputs(H.text.c_str(), sam); } return sam != nullptr; }
		std::vector<psvr::BamRef> refs;
		for (size_t i = 0; i < H.names.size(); ++i) refs.push_back({H.names[i], H.lens[i]});
		return bam.open(fn.c_str(), H.text, refs, threads, level);
	}
	// formatted records (SAM lines or encoded BAM records) of a run of pairs, in order
	void write_raw(const psvr::Bytes &b) { if (b.empty()) return; if (is_bam) bam.write_raw(b.data(), b.size()); else fwrite(b.data(), 1, b.size(), sam); }
	bool close() { if (is_bam) return bam.close(); return fclose(sam) == 0; }
};

struct IndexSvNames : SvNames {
	const psvr_index_t *idx = nullptr;
	const char *print_string(int sv) const override { return psvr_index_sv_print_string(idx, sv); }
	const char *vcf_id(int sv) const override { return psvr_index_sv_vcf_id(idx, sv); }
};

// `panSVR index [-k 22] [--sparse-hash] <anchors.fa> <IndexDir>`: what `deBGA index -k 22 <anchors.fa> <IndexDir>` builds
// (panSVR_run.sh runs it on the SV anchor reference before `aln`).  Host only.
static int index_main(int argc, char **argv)
{
	bool dense = true;
	std::vector<std::string> pos;
	for (int i = 2; i < argc; ++i) {
		if (!strcmp(argv[i], "-k") && i + 1 < argc) { if (atoi(argv[++i]) != 22) { fprintf(stderr, "panSVR aln probes a k = 22 index: -k must be 22\n"); return 1; } }
		else if (!strcmp(argv[i], "--sparse-hash")) dense = false;
		else pos.push_back(argv[i]);
	}
	if (pos.size() != 2) { fprintf(stderr, "usage: panSVR index [-k 22] [--sparse-hash] <anchors.fa> <IndexDir>\n"); return 1; }
	psvr::IndexBuilder b;
	psvr::BuiltIndex ix;
	if (!b.build(pos[0].c_str(), &ix)) { fprintf(stderr, "[panSVR-amd] index: %s\n", b.error().c_str()); return 2; }
	std::string dir = pos[1], err;
	while (dir.size() > 1 && dir.back() == '/') dir.pop_back();
	if (!psvr::IndexBuilder::write_dir(ix, dir, dense, &err)) { fprintf(stderr, "[panSVR-amd] index: %s\n", err.c_str()); return 2; }
	fprintf(stderr, "[panSVR-amd] index: %zu unipaths, %llu distinct 22-mers, %zu positions\n", ix.seqf.size() - 1, (unsigned long long)ix.n_kmer, ix.pos.size());
	return 0;
}

static bool parse_devices(const char *s, std::vector<int> *out)
{
	out->clear();
	const char *p = s;
	while (*p) {
		char *e;
		long a = strtol(p, &e, 10), b = a;
		if (e == p || a < 0) return false;
		if (*e == '-') { const char *q = e + 1; b = strtol(q, &e, 10); if (e == q || b < a) return false; }
		for (long d = a; d <= b; ++d) out->push_back((int)d);
		if (*e == ',') ++e; else if (*e) return false;
		p = e;
	}
	return !out->empty() && out->size() <= 64;
}

[[noreturn]] static void die(const char *what)
{
	fprintf(stderr, "[panSVR-amd] %s: %s\n", what, psvr_last_error());
	abort();                                                // the reference's xassert / xopen end the same way
}

int main(int argc, char **argv)
{
	if (argc >= 2 && !strcmp(argv[1], "index")) return index_main(argc, argv);
	if (argc >= 2 && (!strcmp(argv[1], "signal") || !strcmp(argv[1], "fc_signal"))) return psvr::signal_main(argc, argv);
	if (argc >= 2 && !strcmp(argv[1], "sort")) return psvr::bam_sort_main(argc, argv);
	if (argc < 2 || (strcmp(argv[1], "aln") && strcmp(argv[1], "fc_aln"))) {
		fprintf(stderr, "panSVR (MI355X engine): the read re-alignment step and its two neighbours.\n  usage: panSVR aln|fc_aln [options] <IndexDir> <reads.fq|-> <header.sam>\n         panSVR index [-k 22] <anchors.fa> <IndexDir>\n         panSVR signal [-N] [options] <in.bam> > reads.fq\n         panSVR sort [-n] [-t threads] [-o out.bam] in.bam      (coordinate order + .bai, or -n name order)\n");
		return 1;
	}
	Opt o;
	static struct option lo[] = {{"thread", 1, 0, 't'}, {"gap-open1", 1, 0, 'O'}, {"gap-open2", 1, 0, 'P'}, {"gap-extension1", 1, 0, 'E'}, {"gap-extension2", 1, 0, 'F'},
	                             {"match-score", 1, 0, 'M'}, {"mis-score", 1, 0, 'm'}, {"zdrop", 1, 0, 'z'}, {"band-width", 1, 0, 'w'}, {"output", 1, 0, 'o'},
	                             {"output_signal_ori", 1, 0, 'p'}, {"not-ori", 0, 0, 'Q'}, {"SAM", 0, 0, 'S'}, {"max_use_read", 1, 0, 'R'}, {"device", 1, 0, 1000},
	                             {"records", 1, 0, 1001}, {"trace", 0, 0, 1002}, {"batch", 1, 0, 1003}, {"devices", 1, 0, 1004}, {"batch-bases", 1, 0, 1005}, {"compress-level", 1, 0, 1006}, {"sub-batch", 1, 0, 1007}, {"bgzf-device", 0, 0, 1008}, {"bgzf-fast", 0, 0, 1009},
	                             {"not-use-filter", 0, 0, 'D'}, {"discard-full-match", 0, 0, 'U'}, {"sort-by-name", 0, 0, 'N'}, {0, 0, 0, 0}};
	int c;
	bool sig_by_name = false;
	optind = 2;
	while ((c = getopt_long(argc, argv, "t:O:P:E:F:M:m:z:w:o:p:QSR:DUN", lo, NULL)) >= 0) {
		switch (c) {
		case 't': o.thread_n = atoi(optarg); break;
		case 'O': o.gap_open = atoi(optarg); break;
		case 'P': o.gap_open2 = atoi(optarg); break;
		case 'E': o.gap_ex = atoi(optarg); break;
		case 'F': o.gap_ex2 = atoi(optarg); break;
		case 'M': o.match = atoi(optarg); break;
		case 'm': o.mismatch = atoi(optarg); break;
		case 'z': o.zdrop = atoi(optarg); break;
		case 'w': o.bw = atoi(optarg); break;
		case 'o': o.out = optarg; break;
		case 'p': o.out_ori = optarg; break;
		case 'Q': o.not_ori = true; break;
		case 'S': o.sam = true; break;
		case 'R': o.max_use_read = atoll(optarg); break;
		case 1000: o.devices = {atoi(optarg)}; break;
		case 1001: o.records = optarg; break;
		case 1002: o.trace = true; break;
		case 1003: o.batch_pairs = atoll(optarg); break;
		case 1004: if (!parse_devices(optarg, &o.devices)) { fprintf(stderr, "bad --devices list '%s'\n", optarg); return 1; } break;
		case 1005: o.batch_bases = atoll(optarg); break;
		case 1007: o.sub_pairs = atoll(optarg); break;
		case 1008: o.bgzf_device = true; break;
		case 1009: o.bam_level = psvr::BgzfWriter::kLevelFast; break;
		case 1006: o.bam_level = atoi(optarg); if (o.bam_level < -1 || o.bam_level > 9) { fprintf(stderr, "--compress-level wants -1 .. 9\n"); return 1; } break;
		case 'D': o.sig_all = true; break;
		case 'U': o.sig_discard = true; break;
		case 'N': sig_by_name = true; break;
		default: return usage();
		}
	}
	if (argc - optind < 3) return usage();
	if (!(o.thread_n >= 1 && o.thread_n <= 48)) { fprintf(stderr, "Input error: thread_n cannot be less than 1 or more than 48\n"); abort(); }   // xassert, rr.hpp:121
	if (o.batch_pairs < 1) o.batch_pairs = 1;
	o.index_dir = argv[optind], o.reads = argv[optind + 1], o.header = argv[optind + 2];

	// <reads> may be a BAM (*.bam): the signal step then runs in this process (options of fc_signal: -N for name-sorted input,
	// position-sorted otherwise) and hands its FASTQ text through a pipe to the reader below; <header.sam> is WRITTEN from the BAM's header
	const bool from_bam = o.reads.size() > 4 && o.reads.compare(o.reads.size() - 4, 4, ".bam") == 0;
	if (from_bam) {
		psvr::BamReader rd;
		if (!rd.open(o.reads.c_str())) { fprintf(stderr, "[panSVR-amd] %s\n", rd.error().c_str()); abort(); }
		FILE *h = fopen(o.header.c_str(), "w");
		if (!h) { fprintf(stderr, "fail to open file '%s'\n", o.header.c_str()); abort(); }
		fwrite(rd.header_text.data(), 1, rd.header_text.size(), h);
		fclose(h);
	}
	HeaderInfo H;
	fprintf(stderr, "Open original header file [%s]\n", o.header.c_str());
	if (!H.load(o.header)) { fprintf(stderr, "fail to open file '%s'\n", o.header.c_str()); abort(); }
	fprintf(stderr, "Begin loading index @%s\n", o.index_dir.c_str());
	const double t_idx0 = walltime();
	const int D = (int)o.devices.size();
	for (int d = 0; d < D; ++d) (void)psvr_device_warmup(o.devices[(size_t)d], 4);     // the engines' queues are set up while the index loads
	// one index per DISTINCT device: the first comes from the files, the others from it, device to device
	std::vector<psvr_index_t *> idx((size_t)D, nullptr);
	double t_idx_first = 0, t_idx_clone = 0;
	for (int d = 0; d < D; ++d) {
		int same = -1;
		for (int q = 0; q < d; ++q) if (o.devices[(size_t)q] == o.devices[(size_t)d]) { same = q; break; }
		if (same >= 0) { idx[(size_t)d] = idx[(size_t)same]; continue; }
		const double t0 = walltime();
		if (d == 0) {
			// <IndexDir> may also be the anchor FASTA itself: the index is then built straight into HBM (no `index` step, no files)
			struct stat st;
			const bool is_fasta = stat(o.index_dir.c_str(), &st) == 0 && S_ISREG(st.st_mode);
			if (is_fasta ? psvr_index_build(o.index_dir.c_str(), o.header.c_str(), o.devices[0], &idx[0]) : psvr_index_load(o.index_dir.c_str(), o.header.c_str(), o.devices[0], &idx[0])) die("index");
			t_idx_first = walltime() - t0;
		}
		else { if (psvr_index_clone(idx[0], o.devices[(size_t)d], &idx[(size_t)d])) die("index clone"); t_idx_clone += walltime() - t0; }
	}
	const double t_index = walltime() - t_idx0;
	fprintf(stderr, "End loading index\n");

	fprintf(stderr, "Start classify\n");
	double cpu0 = cputime();
	const double wall0 = walltime();
	std::string fq_path = o.reads;
	psvr::SignalStep sig;
	std::thread sig_thread;
	int sig_rc = 0;
	PairFeed feed;
	if (from_bam) {
		// f2 fused: the signal step's thread hands its pairs straight to the batch being built (PairFeed, fastq_batch.h) -- no FASTQ text, no pipe
		sig.o.sort_by_name = sig_by_name, sig.o.input = o.reads, sig.o.header_fn = o.header, sig.o.status_fn = o.header + ".status";
		sig.o.not_use_filter = o.sig_all, sig.o.discard_full_match = o.sig_discard;
		sig.o.match = o.match, sig.o.mismatch = o.mismatch, sig.o.gap_open = o.gap_open, sig.o.gap_ex = o.gap_ex, sig.o.gap_open2 = o.gap_open2, sig.o.gap_ex2 = o.gap_ex2;
		sig.feed = &feed;
		sig_thread = std::thread([&sig, &sig_rc, &feed]() { sig_rc = sig.run(); feed.close(); });
	}
	FastqReader fq;
	if (from_bam) fq.open_feed(&feed);
	else if (!fq.open(fq_path.c_str())) { fprintf(stderr, "%s\n", fq.error().c_str()); abort(); }
	OutFile fo, fo_ori;
	if (!fo.open(o.out, !o.sam, H, o.thread_n, o.bam_level) || !fo_ori.open(o.out_ori, !o.sam, H, o.thread_n, o.bam_level)) { fprintf(stderr, "fail to open output file\n"); abort(); }
	if (o.bgzf_device && !o.sam) fo.bam.set_device(o.devices[0]), fo_ori.bam.set_device(o.devices[0]);
	FILE *frec = o.records.empty() ? nullptr : fopen(o.records.c_str(), "w");
	fprintf(stderr, "Processing file: [%s].\n", o.reads.c_str());

	psvr_aln_params_t par;
	psvr_aln_params_default(&par);
	par.match = o.match, par.mismatch = o.mismatch, par.gap_open = o.gap_open, par.gap_ex = o.gap_ex, par.gap_open2 = o.gap_open2, par.gap_ex2 = o.gap_ex2, par.zdrop = o.zdrop;
	std::vector<psvr_engine_t *> eng((size_t)D, nullptr);
	// classify_pipeline's overlapped steps (rr.cpp:100-131, kt_pipeline): load_reads | align | output_results -- the last one as two
	// stages here, format | write, because formatting (or BGZF deflate) and the file write each take about as long as the parse.  Job
	// slots cycle through the stages in input order, so the output order is the input order.  A slot keeps its buffers (like the
	// reference's Classify_buff_pool): the raw text + line index of its batch, the page-locked upload arrays, per device the
	// page-locked compact results, and the formatted records.
	struct Block {                       // the share of one device
		long long lo = 0, hi = 0;
		HostBuf hdr_buf, pair_buf, cand_buf, cig_buf;
		ResultView V;
		std::vector<psvr_read_result_t> full; std::vector<uint32_t> full_cig;   // --records only (the fixed-size ABI form)
	};
	struct Job {
		FastqBatch fb;
		std::vector<Block> blk;
		long long pair_base = 0;
		std::vector<psvr::Bytes> mb, ob;   // formatted records of both files, per chunk of pairs
		int state = 0;              // 0 free, 1 loaded, 2 aligned, 3 formatted
		bool last = false;          // end-of-input marker travelling through the stages
		long long batch_pairs_done = 0;   // > 0 on the last piece of a reference-sized batch: that batch's pairs (the progress line)
	};
	// PSVR_CLI_TIMING: when each stage had each piece (ms from the first FASTQ byte), printed at the end
	static const bool cli_tl = getenv("PSVR_CLI_TIMING") != nullptr;
	struct Span { double a = 0, b = 0; };
	std::vector<Span> tl[4];
	if (cli_tl) for (auto &v : tl) v.resize(1 << 16);
	auto mark = [&](int stage, long long piece, double a, double b) { if (cli_tl && piece < (1 << 16)) tl[stage][(size_t)piece].a = a, tl[stage][(size_t)piece].b = b; };
	const int kSlots = 5;
	Job jobs[kSlots];
	for (Job &J : jobs) J.blk = std::vector<Block>((size_t)D);
	std::mutex mu;
	std::condition_variable cv;
	auto wait_state = [&](Job &J, int st) { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return J.state == st; }); };
	auto set_state = [&](Job &J, int st) { { std::lock_guard<std::mutex> lk(mu); J.state = st; } cv.notify_all(); };
	int block = 0;
	double t_read = 0, t_engine = 0, t_format = 0, t_write = 0, t_exchange = 0;
	long long n_batches = 0, n_ref_batches = 0, total_pairs = 0, rebase_iters = 0, d2h_bytes = 0;   // pieces run by the engine; reference-sized batches
	size_t hbm_first = 0, hbm_last = 0;
	EmitStats emit_stats;
	std::thread reader([&]() {
		long long loaded = 0, pair_base = 0, n_read_pieces = 0;
		const long long kFirstPiece = 8192;
		// the reference's batch: N_NEEDED pairs or MAX_read_size bases, whichever comes first (rr.cpp:24,109,126); it is read in pieces
		// that end where it ends (a piece stops at what is left of both limits), so the batches are the reference's
		long long in_batch_pairs = 0, in_batch_bases = 0;
		for (int slot = 0;; slot = (slot + 1) % kSlots) {
			Job &J = jobs[slot];
			wait_state(J, 0);
			long long want = o.batch_pairs - in_batch_pairs;
			if (o.sub_pairs > 0 && o.sub_pairs < want) want = o.sub_pairs;
			// the first pieces are short ones: the later stages have something to do after a millisecond of reading instead of ten, and the
			// engine's first batch -- mostly set-up that does not depend on its size -- is through sooner
			if (o.sub_pairs > 0 && n_read_pieces < 3 && (kFirstPiece << n_read_pieces) < want) want = kFirstPiece << n_read_pieces;
			if (o.max_use_read - loaded < want) want = o.max_use_read - loaded;
			double tw = walltime();
			const bool ok = want > 0 && fq.read(J.fb, want, o.batch_bases - in_batch_bases, o.thread_n);
			t_read += walltime() - tw;
			mark(0, n_read_pieces++, tw, walltime());
			if (!ok) { J.last = true; J.batch_pairs_done = in_batch_pairs; set_state(J, 1); return; }
			if (loaded == 0) fq.stat_params(&par);          // STAT_ of the very first read (rr.cpp:134-148), before the first batch is aligned
			loaded += J.fb.n_pairs();
			J.pair_base = pair_base, pair_base += J.fb.n_pairs();
			in_batch_pairs += J.fb.n_pairs(), in_batch_bases += J.fb.base_off[J.fb.R];
			J.batch_pairs_done = 0;
			if (in_batch_pairs >= o.batch_pairs || in_batch_bases >= o.batch_bases) J.batch_pairs_done = in_batch_pairs, in_batch_pairs = in_batch_bases = 0;
			set_state(J, 1);
		}
	});
	IndexSvNames svn;
	svn.idx = idx[0];
	SamEmitter em;
	em.H = &H, em.sv = &svn, em.as_bam = !o.sam, em.not_ori = o.not_ori, em.stats = &emit_stats;
	std::thread formatter([&]() {
		long long n_fmt_pieces = 0;
		for (int slot = 0;; slot = (slot + 1) % kSlots) {
			Job &J = jobs[slot];
			wait_state(J, 2);
			if (J.last) {
				if (J.batch_pairs_done > 0) fprintf(stderr, "Processing %d reads, at block ID %d\n", (int)J.batch_pairs_done, block++), ++n_ref_batches;   // the input ended inside a batch
				set_state(J, 3);
				return;
			}
			const long long P = J.fb.n_pairs();
			double tw = walltime();
			if (J.batch_pairs_done > 0) fprintf(stderr, "Processing %d reads, at block ID %d\n", (int)J.batch_pairs_done, block++), ++n_ref_batches;     // output_results, rr.cpp:166
			em.min_filter_score = par.min_filter_score;
			if (frec) {
				for (const Block &bk : J.blk)
					for (long long p = bk.lo; p < bk.hi; ++p) {
						const char *t; int lens[2];
						J.fb.seq(2 * p, t, lens[0]), J.fb.seq(2 * p + 1, t, lens[1]);
						fprintf(frec, "%s\n", record_json(J.pair_base + p, &bk.full[(size_t)(2 * (p - bk.lo))], bk.V.pairs[p - bk.lo], &J.fb.ori[2 * p], lens, bk.full_cig.data(), o.trace).c_str());
					}
			}
			// ---- step 2: records of both files, formatted for runs of pairs on -t threads and written in input order
			const long long chunk = 4096, nchunk = (P + chunk - 1) / chunk;
			std::vector<psvr::Bytes> &mb = J.mb, &ob = J.ob;
			mb.resize((size_t)nchunk), ob.resize((size_t)nchunk);
			for (auto &v : mb) v.clear();        // (capacity is kept from the slot's previous batch: no growth copies in steady state)
			for (auto &v : ob) v.clear();
			std::atomic<long long> next(0);
			auto work = [&]() {
				for (long long ci = next++; ci < nchunk; ci = next++) {
					const long long p0 = ci * chunk, p1 = p0 + chunk < P ? p0 + chunk : P;
					size_t bi = 0;
					for (long long p = p0; p < p1; ++p) {
						while (p >= J.blk[bi].hi) ++bi;
						em.main_pair(J.fb, J.blk[bi].V, p, mb[(size_t)ci]), em.ori_pair(J.fb, J.blk[bi].V, p, ob[(size_t)ci]);
					}
				}
			};
			thread_pool().run((int)(o.thread_n < nchunk ? o.thread_n : nchunk), [&](int) { work(); });
			t_format += walltime() - tw;
			mark(2, n_fmt_pieces++, tw, walltime());
			set_state(J, 3);
		}
	});
	std::thread writer([&]() {
		long long n_wr_pieces = 0;
		for (int slot = 0;; slot = (slot + 1) % kSlots) {
			Job &J = jobs[slot];
			wait_state(J, 3);
			if (J.last) return;
			const double tw = walltime();
			for (size_t ci = 0; ci < J.mb.size(); ++ci) fo.write_raw(J.mb[ci]), fo_ori.write_raw(J.ob[ci]);
			t_write += walltime() - tw;
			mark(3, n_wr_pieces++, tw, walltime());
			set_state(J, 0);
		}
	});
	// ---- step 1: the engine(s)
	int64_t pos[3] = {0, 0, 0};                          // where the next batch starts in the three draw streams
	for (int slot = 0;; slot = (slot + 1) % kSlots) {
		Job &J = jobs[slot];
		wait_state(J, 1);
		if (J.last) { set_state(J, 2); break; }
		double tw = walltime();
		if (!eng[0]) {
			fprintf(stderr, "Current used read status: READ_LEN=%d; ISIZE_MIN=%d; ISIZE_MID=%d; ISIZE_MAX=%d; filter_score_full_match=%d\n", par.normal_read_length, par.isize_min, 0,
			        par.isize_max, par.min_filter_score);
			for (int d = 0; d < D; ++d) if (psvr_engine_create(idx[(size_t)d], &par, &eng[(size_t)d])) die("engine");
			if (psvr_engine_stream_end(eng[0], pos)) die("engine");      // a fresh engine stands where the reference's generators stand after init_run
		}
		const long long P = J.fb.n_pairs();
		for (int d = 0; d < D; ++d) { J.blk[(size_t)d].lo = (P * d + D - 1) / D, J.blk[(size_t)d].hi = (P * (d + 1) + D - 1) / D; }
		auto each_device = [&](auto &&fn) {                 // one host thread per device (the engine API is single-owner per engine)
			std::vector<std::thread> th;
			for (int d = 1; d < D; ++d) th.emplace_back(fn, d);
			fn(0);
			for (std::thread &t : th) t.join();
		};
		std::vector<int> rcs((size_t)D, 0);
		std::vector<std::string> errs((size_t)D);
		auto fail_check = [&]() { for (int d = 0; d < D; ++d) if (rcs[(size_t)d]) { fprintf(stderr, "[panSVR-amd] engine error %d on device %d: %s\n", rcs[(size_t)d], o.devices[(size_t)d], errs[(size_t)d].c_str()); abort(); } };
		static const bool cli_timing = getenv("PSVR_CLI_TIMING") != nullptr;
		each_device([&](int d) {
			Block &bk = J.blk[(size_t)d];
			const long long n = bk.hi - bk.lo;
			const double t0 = walltime();
			int rc = psvr_engine_set_stream_pos(eng[(size_t)d], pos);        // block 0 starts there; the others are moved below
			if (!rc) rc = psvr_engine_upload(eng[(size_t)d], n, J.fb.bases, J.fb.base_off + 2 * bk.lo, J.fb.ori + 2 * bk.lo);
			const double t1 = walltime();
			if (!rc) rc = psvr_engine_run(eng[(size_t)d], o.trace ? 1 : 0, nullptr);
			if (cli_timing && d == 0) fprintf(stderr, "[panSVR-amd] batch %lld: engine ready %.1f ms after the batch, upload %.1f ms, run %.1f ms\n", n_batches, (t0 - tw) * 1e3, (t1 - t0) * 1e3, (walltime() - t1) * 1e3);
			if (rc) rcs[(size_t)d] = rc, errs[(size_t)d] = psvr_last_error();
		});
		fail_check();
		// the draw-order exchange: block d starts where block d-1 ended.  A block's draw count almost never depends on where it
		// starts, so one pass of moves normally settles it; the loop covers the rest.
		if (D > 1) {
			const double tx = walltime();
			std::vector<int64_t> start((size_t)D * 3), end((size_t)D * 3);
			for (int d = 0; d < D; ++d) for (int k = 0; k < 3; ++k) start[(size_t)d * 3 + k] = pos[k];
			for (int it = 0;; ++it) {
				for (int d = 0; d < D; ++d) if (psvr_engine_stream_end(eng[(size_t)d], &end[(size_t)d * 3])) die("engine");
				std::vector<int> moved;
				int64_t acc[3] = {pos[0], pos[1], pos[2]};
				for (int d = 0; d < D; ++d) {
					int64_t used[3];
					for (int k = 0; k < 3; ++k) used[k] = end[(size_t)d * 3 + k] - start[(size_t)d * 3 + k];
					bool mv = false;
					for (int k = 0; k < 3; ++k) if (start[(size_t)d * 3 + k] != acc[k]) mv = true, start[(size_t)d * 3 + k] = acc[k];
					if (mv) moved.push_back(d);
					for (int k = 0; k < 3; ++k) acc[k] += used[k];
				}
				if (moved.empty()) break;
				if (it > 64) { fprintf(stderr, "[panSVR-amd] draw-order exchange did not converge\n"); abort(); }
				std::vector<std::thread> th;
				for (int d : moved) th.emplace_back([&, d]() { if (psvr_engine_rebase(eng[(size_t)d], &start[(size_t)d * 3], nullptr)) rcs[(size_t)d] = 1, errs[(size_t)d] = psvr_last_error(); });
				for (std::thread &t : th) t.join();
				fail_check();
				++rebase_iters;
			}
			t_exchange += walltime() - tx;
		}
		if (psvr_engine_stream_end(eng[(size_t)D - 1], pos)) die("engine");
		each_device([&](int d) {
			Block &bk = J.blk[(size_t)d];
			const long long n = bk.hi - bk.lo;
			int64_t nc = 0, nw = 0;
			int rc = psvr_engine_download_compact(eng[(size_t)d], nullptr, nullptr, nullptr, 0, &nc, nullptr, 0, &nw);
			if (rc == PSVR_ERR_OVERFLOW) rc = 0;
			psvr_read_hdr_t *hdr = (psvr_read_hdr_t *)bk.hdr_buf.reserve((size_t)(2 * n + 1) * sizeof(psvr_read_hdr_t));
			psvr_pair_result_t *prs = (psvr_pair_result_t *)bk.pair_buf.reserve((size_t)(n + 1) * sizeof(psvr_pair_result_t));
			psvr_cand_t *cands = (psvr_cand_t *)bk.cand_buf.reserve((size_t)(nc + 1) * sizeof(psvr_cand_t));
			uint32_t *cig = (uint32_t *)bk.cig_buf.reserve((size_t)(nw + 1) * 4);
			if (!rc) rc = psvr_engine_download_compact(eng[(size_t)d], hdr, prs, cands, nc + 1, &nc, cig, nw + 1, &nw);
			bk.V.hdr = hdr, bk.V.pairs = prs, bk.V.cands = cands, bk.V.cig = cig, bk.V.pair0 = bk.lo;
			if (!rc && frec) {                               // the parity tests read the fixed-size ABI records
				int64_t used = 0;
				rc = psvr_engine_download(eng[(size_t)d], nullptr, nullptr, nullptr, 0, &used);
				if (rc == PSVR_ERR_OVERFLOW) rc = 0;
				bk.full.resize((size_t)(2 * n)), bk.full_cig.resize((size_t)used + 1);
				if (!rc) rc = psvr_engine_download(eng[(size_t)d], bk.full.data(), nullptr, bk.full_cig.data(), used + 1, &used);
			}
			if (rc) rcs[(size_t)d] = rc, errs[(size_t)d] = psvr_last_error();
			else { std::lock_guard<std::mutex> lk(mu); d2h_bytes += (long long)(2 * n * sizeof(psvr_read_hdr_t) + n * sizeof(psvr_pair_result_t) + nc * sizeof(psvr_cand_t) + nw * 4); }
		});
		fail_check();
		{   // steady footprint: HBM in use on the first device after the first and after the latest batch
			char sb[8192];
			if (!psvr_engine_stats(eng[0], sb, sizeof sb)) { const char *q = strstr(sb, "\"hbm_used_bytes\":"); if (q) { hbm_last = strtoull(q + 17, nullptr, 10); if (n_batches <= 3) hbm_first = hbm_last; } }      // (first: after the first piece of full size -- the three before it are short ones)
		}
		++n_batches, total_pairs += P;
		t_engine += walltime() - tw;
		mark(1, n_batches - 1, tw, walltime());
		set_state(J, 2);
	}
	reader.join(), formatter.join(), writer.join();
	feed.abort();                                        // (a reader that stopped at -R leaves the signal step to run to its end unheard)
	if (sig_thread.joinable()) {
		sig_thread.join();
		if (sig_rc) { fprintf(stderr, "[panSVR-amd] the signal step failed\n"); abort(); }
	}
	if (!fo.close() || !fo_ori.close()) { fprintf(stderr, "fail to write output file\n"); abort(); }
	if (frec) fclose(frec);
	const double wall = walltime() - wall0;              // first FASTQ byte to the files closed; giving the HBM back is reported beside it, like the index load
	for (int d = 0; d < D; ++d) if (eng[(size_t)d]) psvr_engine_destroy(eng[(size_t)d]);
	for (int d = 0; d < D; ++d) {
		bool dup = false;
		for (int q = 0; q < d; ++q) if (idx[(size_t)q] == idx[(size_t)d]) dup = true;
		if (!dup) psvr_index_destroy(idx[(size_t)d]);
	}
	const double t_teardown = walltime() - wall0 - wall;
	if (cli_tl) {
		static const char *nm[4] = {"read", "engine", "format", "write"};
		for (long long i = 0; i <= n_batches && i < (1 << 16); ++i) {
			fprintf(stderr, "[panSVR-amd] piece %lld:", i);
			for (int st = 0; st < 4; ++st) fprintf(stderr, "  %s %.1f-%.1f", nm[st], (tl[st][(size_t)i].a - wall0) * 1e3, (tl[st][(size_t)i].b - wall0) * 1e3);
			fprintf(stderr, "\n");
		}
		fprintf(stderr, "[panSVR-amd] files closed at %.1f ms, engine and index released %.1f ms later\n", wall * 1e3, t_teardown * 1e3);
	}
	fprintf(stderr, "Classify CPU: %.3f sec\n", cputime() - cpu0);
	if (emit_stats.dropped) fprintf(stderr, "[panSVR-amd] %lld records were refused by the record rules of sam_parse1 and not written (see the ERROR lines above)\n", (long long)emit_stats.dropped);
	fprintf(stderr, "[panSVR-amd] wall: read+parse %.3f s, engine (upload+run+download) %.3f s, format %.3f s, write%s %.3f s\n", t_read, t_engine, t_format, o.sam ? "" : "+compress", t_write);
	fprintf(stderr,
	        "[panSVR-amd] e2e_json {\"pairs\":%lld,\"batches\":%lld,\"pieces\":%lld,\"devices\":%d,\"threads\":%d,\"wall_s\":%.4f,\"index_s\":%.4f,\"index_first_s\":%.4f,\"index_clone_s\":%.4f,\"read_parse_s\":%.4f,"
	        "\"engine_s\":%.4f,\"exchange_s\":%.4f,\"rebase_iterations\":%lld,\"format_s\":%.4f,\"write_s\":%.4f,\"d2h_bytes\":%lld,\"hbm_used_first\":%zu,\"hbm_used_last\":%zu,\"dropped\":%lld,\"teardown_s\":%.4f}\n",
	        total_pairs, n_ref_batches, n_batches, D, o.thread_n, wall, t_index, t_idx_first, t_idx_clone, t_read, t_engine, t_exchange, rebase_iters, t_format, t_write, d2h_bytes, hbm_first, hbm_last,
	        (long long)emit_stats.dropped, t_teardown);
	return 0;
}
