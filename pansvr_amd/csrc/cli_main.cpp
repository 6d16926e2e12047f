// cli_main.cpp -- `panSVR aln` / `panSVR fc_aln` on the MI355X engine: the host side of the reference's
// three-stage pipeline (load_reads -> [engine] -> output_results; src/PanSVgenerateVCF/read_realignment.cpp:26-176)
// above the C ABI of include/psvr_engine.h.  Same options, positional arguments, stderr progress
// lines and SAM records as the reference; every other sub-command of panSVR is out of scope.
//
// It links libpsvr_engine.so only through psvr_engine.h; host_io.h is host-side parsing/formatting.
#include <getopt.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>
#include <time.h>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <signal.h>
#include <unistd.h>
#include <vector>
#include "../../include/psvr_engine.h"
#include "host_io.h"
#include "bam_writer.h"
#include "index_build.h"
#include "signal_step.h"

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
	int device = 0;
	long long batch_pairs = 2000000;   // N_NEEDED, rr.cpp:24
	bool sig_all = false, sig_discard = false;   // BAM input: fc_signal's -D / -U
};

static int usage()
{
	fprintf(stderr,
	        "\n  Usage:     panSVR  aln|fc_aln  [Options] <IndexDir> [ReadFiles.fa][ori_header_fn.sam]>\n"
	        "  Basic:   \n"
	        "    <IndexDir>      FOLDER   the directory contains index\n"
	        "    [ReadFiles.fa]  FILES    reads files, FASTQ(A) format, read 1 and 2 of a pair stored together ('-' = stdin),\n"
	        "                             or a name-sorted *.bam: the signal step then runs in-process ([ori_header.sam] is written;\n"
	        "                             -D / -U as in fc_signal: all pairs are signals / drop fully matching pairs)\n"
	        "                             Using [signal] command to generate this type of file\n"
	        "    [ori_header.sam]  FILES  Header file of original BAM/CRAM file\n"
	        "  Options:\n"
	        "    -t, --thread            INT  accepted for compatibility (the engine runs on the GPU) [4]\n"
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
	        "        --device            INT  HIP device [0]\n"
	        "        --records           STR  dump per-pair decision records (JSON lines) for parity checks\n"
	        "        --trace                  add per-strand seed/chain hashes to --records\n\n");
	return 1;
}

static double cputime()
{
	return (double)clock() / CLOCKS_PER_SEC;
}

static double walltime()
{
	struct timeval tv;
	gettimeofday(&tv, nullptr);
	return tv.tv_sec + 1e-6 * tv.tv_usec;
}

static char rc_char(char c)   // getReverseChar, clib/bam_file.c:316-327
{
	switch (c) {
	case 'A': case 'a': return 'T';
	case 'C': case 'c': return 'G';
	case 'G': case 'g': return 'C';
	case 'T': case 't': return 'A';
	}
	return 'N';
}
static void rev_seq(std::string &s)   // getReverseStr_char, clib/bam_file.c:329-339
{
	int len = (int)s.size(), half = len >> 1;
	for (int i = 0; i < half; i++) { char t = s[i]; s[i] = rc_char(s[len - 1 - i]); s[len - 1 - i] = rc_char(t); }
	if (len & 1) s[half] = rc_char(s[half]);
}
static void rev_qual(std::string &q)  // getReverseStr_qual_char, clib/bam_file.c:351-359: loop bound len/2 + 1 (even len: middle pair swapped back)
{
	int len = (int)q.size(), half = len >> 1;
	for (int i = 0; i < half + 1; i++) { int ri = len - 1 - i; if (ri < 0 || i >= len) break; char t = q[i]; q[i] = q[ri]; q[ri] = t; }
}

struct HeaderInfo {
	std::string text;
	std::vector<std::string> names;
	std::vector<uint32_t> lens;
	const char *name(int id) const { return id >= 0 && id < (int)names.size() ? names[id].c_str() : "*"; }
};

static bool load_header(const std::string &fn, HeaderInfo *h)
{
	FILE *f = fopen(fn.c_str(), "r");
	if (!f) return false;
	char buf[65536];
	while (fgets(buf, sizeof buf, f)) {
		if (buf[0] != '@') continue;
		h->text += buf;
		if (strncmp(buf, "@SQ", 3)) continue;
		char *p = strstr(buf, "SN:");
		if (!p) continue;
		p += 3;
		char *e = p;
		while (*e && *e != '\t' && *e != '\n') ++e;
		h->names.emplace_back(p, e - p);
		const char *ln = strstr(buf, "LN:");
		h->lens.push_back(ln ? (uint32_t)strtoul(ln + 3, nullptr, 10) : 0u);
	}
	fclose(f);
	return true;
}

// one output file: SAM text (-S) or BAM (default, like the reference's init_run)
struct OutFile {
	FILE *sam = nullptr;
	psvr::BamWriter bam;
	bool is_bam = false;
	bool open(const std::string &fn, bool as_bam, const HeaderInfo &H, int threads)
	{
		is_bam = as_bam;
		if (!as_bam) { sam = fopen(fn.c_str(), "w"); if (sam) fputs(H.text.c_str(), sam); return sam != nullptr; }
		std::vector<psvr::BamRef> refs;
		for (size_t i = 0; i < H.names.size(); ++i) refs.push_back({H.names[i], H.lens[i]});
		return bam.open(fn.c_str(), H.text, refs, threads);
	}
	// formatted records (SAM lines or encoded BAM records) of a run of pairs, in order
	void write_raw(const std::vector<uint8_t> &b) { if (b.empty()) return; if (is_bam) bam.write_raw(b.data(), b.size()); else fwrite(b.data(), 1, b.size(), sam); }
	bool close() { if (is_bam) return bam.close(); return fclose(sam) == 0; }
};

// what survives sam_parse1 -> sam_write1 (htslib 1.9 sam.c:1197-1424, sam_format1) for the text built by
// single_end_handler::output_BAM (rr.cpp:479-536): POS <= 0 drops the record, RNEXT collapses to '=' ...
static bool emit_record(const OutFile &out, std::vector<uint8_t> &dst, const HeaderInfo &H, const std::string &name, int flag, int chr_id, uint32_t ref_bg, int mapq, const std::string &cigar,
                        bool has_mate, int mate_chr, uint32_t mate_pos, int isize, const std::string &seq, const std::string &qual, const std::string &tags)
{
	int pos = (int)ref_bg;                               // printed with %d
	if (chr_id < 0 || chr_id >= (int)H.names.size()) return false;   // target_name[] would be out of range in the reference
	if (pos - 1 < 0) return false;                       // "mapped query cannot have zero coordinate; treated as unmapped" -> tid = -1 -> not written
	std::string rnext = "*";
	long pnext = 0;
	int mtid = -1;
	if (has_mate) {
		int mp = (int)mate_pos;
		bool mate_ok = mate_chr >= 0 && mate_chr < (int)H.names.size() && !(mp - 1 < 0);
		if (mate_ok) rnext = mate_chr == chr_id ? "=" : H.name(mate_chr), mtid = mate_chr;
		pnext = mp;
	}
	if (out.is_bam) {
		psvr::SamFields f;
		f.qname = name, f.flag = flag, f.tid = chr_id, f.pos1 = pos, f.mapq = mapq, f.cigar = cigar.empty() ? "*" : cigar;
		f.mtid = mtid, f.mpos1 = pnext, f.isize = isize, f.seq = seq, f.qual = qual, f.tags = tags;
		return psvr::BamWriter::encode(f, dst);
	}
	char head[512];
	int n = snprintf(head, sizeof head, "\t%d\t%s\t%d\t%d\t", flag, H.name(chr_id), pos, mapq);
	auto put = [&](const char *p, size_t m) { dst.insert(dst.end(), (const uint8_t *)p, (const uint8_t *)p + m); };
	put(name.data(), name.size()), put(head, (size_t)n);
	if (cigar.empty()) put("*", 1); else put(cigar.data(), cigar.size());
	n = snprintf(head, sizeof head, "\t%s\t%ld\t%d\t", rnext.c_str(), pnext, isize);
	put(head, (size_t)n), put(seq.data(), seq.size()), put("\t", 1), put(qual.data(), qual.size()), put(tags.data(), tags.size()), put("\n", 1);
	return true;
}

// single_end_handler::output_ori_bam (rr.cpp:656-717): the ORIGINAL alignment re-assembled from the comment's
// FLAG_/CIGAR_/MATE_/TAG_ sections (+ MS:i:max_score); returns false when the record would not be written
struct OriRecord { int flag = 0, mapq = 0, mate_chr = -1, mate_pos = 0, isize = 0; std::string cigar, tags; };
static bool parse_ori_record(const std::string &comment, OriRecord *r)
{
	const char *c = comment.c_str();
	const char *f = strstr(c, "FLAG_");
	if (!f) return false;
	unsigned fl = 0, q = 0;
	if (sscanf(f + 5, "%u_%u_", &fl, &q) < 2) return false;
	r->flag = (int)fl, r->mapq = (int)q;
	const char *cg = strstr(f + 5, "CIGAR_");
	if (!cg) return false;
	cg += 6;
	const char *ce = strchr(cg, '_');
	if (!ce) return false;
	r->cigar.assign(cg, ce - cg);
	const char *mate = ce + 1 + 5;                           // skips "MATE_"
	if (strlen(ce) < 6 || sscanf(mate, "%d_%d_%d_", &r->mate_chr, &r->mate_pos, &r->isize) < 3) return false;
	r->mate_pos += 1;
	const char *tg = strstr(mate, "TAG_");
	if (!tg) return false;
	std::string tags = tg + 4;
	const int tl = (int)tags.size();
	for (int i = 0; i < tl - 5; i++) if (tags[i] == '_' && tags[i + 3] == ':' && tags[i + 5] == ':') tags[i] = '\t';
	if (tl > 0) tags.resize(tl - 1);
	r->tags = tags;
	return true;
}

// bam_has_clip_or_unmapped_ori (rr.cpp:721-733) on the CIGAR text
static bool ori_has_clip(const std::string &cigar, int min_clip)
{
	if (cigar.empty() || cigar == "*") return true;
	std::vector<std::pair<int, char>> ops;
	int n = 0;
	for (char ch : cigar) { if (ch >= '0' && ch <= '9') n = n * 10 + (ch - '0'); else { ops.push_back({n, ch}); n = 0; } }
	if (ops.empty()) return true;
	int tot = 0;
	if (ops.front().second == 'S' || ops.front().second == 'H') tot += ops.front().first;
	if (ops.back().second == 'S' || ops.back().second == 'H') tot += ops.back().first;
	return tot >= min_clip;
}

static std::string cigar_string(const psvr_cand_t &c, const uint32_t *cig)
{
	std::string s;
	char b[32];
	for (uint32_t j = 0; j < c.n_cigar; ++j) { uint32_t w = cig[c.cigar_off + j]; snprintf(b, sizeof b, "%d%c", (int)(int16_t)(w >> 4), "MIDNSHP=XB"[w & 0xf]); s += b; }
	return s;
}

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

int main(int argc, char **argv)
{
	if (argc >= 2 && !strcmp(argv[1], "index")) return index_main(argc, argv);
	if (argc >= 2 && (!strcmp(argv[1], "signal") || !strcmp(argv[1], "fc_signal"))) return psvr::signal_main(argc, argv);
	if (argc < 2 || (strcmp(argv[1], "aln") && strcmp(argv[1], "fc_aln"))) {
		fprintf(stderr, "panSVR (MI355X engine): the read re-alignment step and its two neighbours.\n  usage: panSVR aln|fc_aln [options] <IndexDir> <reads.fq|-> <header.sam>\n         panSVR index [-k 22] <anchors.fa> <IndexDir>\n         panSVR signal -N [options] <name-sorted.bam> > reads.fq\n");
		return 1;
	}
	Opt o;
	static struct option lo[] = {{"thread", 1, 0, 't'}, {"gap-open1", 1, 0, 'O'}, {"gap-open2", 1, 0, 'P'}, {"gap-extension1", 1, 0, 'E'}, {"gap-extension2", 1, 0, 'F'},
	                             {"match-score", 1, 0, 'M'}, {"mis-score", 1, 0, 'm'}, {"zdrop", 1, 0, 'z'}, {"band-width", 1, 0, 'w'}, {"output", 1, 0, 'o'},
	                             {"output_signal_ori", 1, 0, 'p'}, {"not-ori", 0, 0, 'Q'}, {"SAM", 0, 0, 'S'}, {"max_use_read", 1, 0, 'R'}, {"device", 1, 0, 1000},
	                             {"records", 1, 0, 1001}, {"trace", 0, 0, 1002}, {"batch", 1, 0, 1003}, {"not-use-filter", 0, 0, 'D'}, {"discard-full-match", 0, 0, 'U'}, {0, 0, 0, 0}};
	int c;
	optind = 2;
	while ((c = getopt_long(argc, argv, "t:O:P:E:F:M:m:z:w:o:p:QSR:DU", lo, NULL)) >= 0) {
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
		case 1000: o.device = atoi(optarg); break;
		case 1001: o.records = optarg; break;
		case 1002: o.trace = true; break;
		case 1003: o.batch_pairs = atoll(optarg); break;
		case 'D': o.sig_all = true; break;
		case 'U': o.sig_discard = true; break;
		default: return usage();
		}
	}
	if (argc - optind < 3) return usage();
	if (!(o.thread_n >= 1 && o.thread_n <= 48)) { fprintf(stderr, "Input error: thread_n cannot be less than 1 or more than 48\n"); abort(); }   // xassert, rr.hpp:121
	o.index_dir = argv[optind], o.reads = argv[optind + 1], o.header = argv[optind + 2];

	// <reads> may be a name-sorted BAM (*.bam): the signal step then runs in this process (default options of fc_signal) and its
	// FASTQ goes through a pipe to the reader below; <header.sam> is WRITTEN from the BAM's header in that case
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
	if (!load_header(o.header, &H)) { fprintf(stderr, "fail to open file '%s'\n", o.header.c_str()); abort(); }
	fprintf(stderr, "Begin loading index @%s\n", o.index_dir.c_str());
	psvr_index_t *idx = nullptr;
	if (psvr_index_load(o.index_dir.c_str(), o.header.c_str(), o.device, &idx)) { fprintf(stderr, "[panSVR-amd] %s\n", psvr_last_error()); abort(); }
	fprintf(stderr, "End loading index\n");

	fprintf(stderr, "Start classify\n");
	double cpu0 = cputime();
	FILE *fq = nullptr;
	psvr::SignalStep sig;
	std::thread sig_thread;
	int sig_rc = 0;
	if (from_bam) {
		signal(SIGPIPE, SIG_IGN);            // if the reader stops early (-R), the signal step's writes fail quietly and it runs to its end
		int fds[2];
		if (pipe(fds)) { fprintf(stderr, "[panSVR-amd] pipe() failed\n"); abort(); }
		sig.o.sort_by_name = true, sig.o.input = o.reads, sig.o.header_fn = o.header, sig.o.status_fn = o.header + ".status";
		sig.o.not_use_filter = o.sig_all, sig.o.discard_full_match = o.sig_discard;
		sig.o.match = o.match, sig.o.mismatch = o.mismatch, sig.o.gap_open = o.gap_open, sig.o.gap_ex = o.gap_ex, sig.o.gap_open2 = o.gap_open2, sig.o.gap_ex2 = o.gap_ex2;
		FILE *w = fdopen(fds[1], "w");
		sig.out = w;
		sig_thread = std::thread([&sig, &sig_rc, w]() { sig_rc = sig.run(); fclose(w); });
		fq = fdopen(fds[0], "r");
	} else fq = o.reads == "-" ? stdin : fopen(o.reads.c_str(), "r");
	if (!fq) { fprintf(stderr, "fail to open file '%s'\n", o.reads.c_str()); abort(); }
	OutFile fo, fo_ori;
	if (!fo.open(o.out, !o.sam, H, o.thread_n) || !fo_ori.open(o.out_ori, !o.sam, H, o.thread_n)) { fprintf(stderr, "fail to open output file\n"); abort(); }
	FILE *frec = o.records.empty() ? nullptr : fopen(o.records.c_str(), "w");
	fprintf(stderr, "Processing file: [%s].\n", o.reads.c_str());

	psvr_aln_params_t par;
	psvr_aln_params_default(&par);
	par.match = o.match, par.mismatch = o.mismatch, par.gap_open = o.gap_open, par.gap_ex = o.gap_ex, par.gap_open2 = o.gap_open2, par.gap_ex2 = o.gap_ex2, par.zdrop = o.zdrop;
	psvr_engine_t *eng = nullptr;
	// classify_pipeline's three overlapped steps (rr.cpp:100-131, kt_pipeline): load_reads | align | output_results.  Three job
	// slots cycle through the stages in input order, so the output order is the input order.
	// The record buffers live as long as their job slot (like the reference's Classify_buff_pool) and are page-locked when the
	// library can provide that: 1.3 GB of fixed-size records per 1 M pairs come back at the link's rate instead of a third of it.
	struct HostBuf {
		void *p = nullptr; size_t cap = 0; bool locked = false;
		void *reserve(size_t bytes)
		{
			if (bytes <= cap) return p;
			release();
			const size_t want = bytes + bytes / 8;
			if ((p = psvr_host_alloc(want))) locked = true;
			else if (!(p = malloc(want))) { fprintf(stderr, "[panSVR-amd] out of host memory\n"); abort(); }
			cap = want;
			return p;
		}
		void release() { if (p) { if (locked) psvr_host_free(p); else free(p); } p = nullptr, cap = 0, locked = false; }
		~HostBuf() { release(); }
	};
	struct Job {
		std::vector<FqRec> recs; std::vector<char> bases; std::vector<long long> base_off; std::vector<psvr_ori_t> ori;
		HostBuf res_buf, cig_buf;
		psvr_read_result_t *res = nullptr; uint32_t *cig = nullptr;
		std::vector<psvr_pair_result_t> pres;
		long long pair_base = 0;
		int state = 0;              // 0 free, 1 loaded, 2 aligned
		bool last = false;          // end-of-input marker travelling through the stages
		long long n_pairs() const { return (long long)recs.size() / 2; }
	};
	Job jobs[3];
	std::mutex mu;
	std::condition_variable cv;
	auto wait_state = [&](Job &J, int st) { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return J.state == st; }); };
	auto set_state = [&](Job &J, int st) { { std::lock_guard<std::mutex> lk(mu); J.state = st; } cv.notify_all(); };
	int block = 0;
	double t_read = 0, t_engine = 0, t_format = 0, t_write = 0;
	std::thread reader([&]() {
		FastqBatch fb;
		long long loaded = 0, pair_base = 0;
		for (int slot = 0;; slot = (slot + 1) % 3) {
			Job &J = jobs[slot];
			wait_state(J, 0);
			long long want = o.batch_pairs;
			if (o.max_use_read - loaded < want) want = o.max_use_read - loaded;
			double tw = walltime();
			const bool ok = want > 0 && fb.read(fq, want, o.thread_n);
			t_read += walltime() - tw;
			if (!ok) { J.last = true; set_state(J, 1); return; }
			if (loaded == 0) fb.stat_params(&par);          // STAT_ of the very first read (rr.cpp:134-148), before the first batch is aligned
			loaded += fb.n_pairs();
			J.recs.swap(fb.recs), J.bases.swap(fb.bases), J.base_off.swap(fb.base_off), J.ori.swap(fb.ori);
			J.pair_base = pair_base, pair_base += J.n_pairs();
			set_state(J, 1);
		}
	});
	std::thread writer([&]() {
		for (int slot = 0;; slot = (slot + 1) % 3) {
			Job &J = jobs[slot];
			wait_state(J, 2);
			if (J.last) return;
			const long long P = J.n_pairs();
		double tw = walltime();
		fprintf(stderr, "Processing %d reads, at block ID %d\n", (int)P, block++);     // output_results, rr.cpp:166
		if (frec) {
			for (long long p = 0; p < P; ++p) {
				int lens[2] = {(int)J.recs[2 * p].seq.size(), (int)J.recs[2 * p + 1].seq.size()};
				fprintf(frec, "%s\n", record_json(J.pair_base + p, &J.res[2 * p], J.pres[p], &J.ori[2 * p], lens, J.cig, o.trace).c_str());
			}
		}
		// ---- step 2: records (output_BAM, rr.cpp:479-536), formatted for runs of pairs on -t threads and written in input order
		auto format_main = [&](long long p0, long long p1, std::vector<uint8_t> &dst) {
		for (long long p = p0; p < p1; ++p) {
			const psvr_pair_result_t &pr = J.pres[p];
			if (!pr.gain) continue;
			for (int k = 0; k < 2; ++k) {
				const psvr_read_result_t &rr = J.res[2 * p + k];
				const FqRec &rec = J.recs[2 * p + k];
				const psvr_ori_t &ori = J.ori[2 * p + k];
				if (rr.primary == -1) continue;                          // primary_result == NULL
				const bool is_ori = rr.primary == -2;
				if (o.not_ori && is_ori) continue;
				int chr_id, direction, mapq;
				uint32_t ref_bg, align_score, chain_score = 0;
				std::string cg;
				char b[256];
				if (is_ori) {
					chr_id = ori.chr_id, direction = ori.direction, mapq = ori.mapq, ref_bg = ori.ref_bg >= 0x7fffffffu ? 1u : ori.ref_bg, align_score = ori.align_score;
					if (ori.read_bg > 0) { snprintf(b, sizeof b, "%dS", (int)(int16_t)(uint16_t)ori.read_bg); cg += b; }
					snprintf(b, sizeof b, "%dM", (int)(int16_t)(uint16_t)((int)rec.seq.size() - (int)ori.read_bg));
					cg += b;
				} else {
					const psvr_cand_t &cd = rr.cand[rr.primary];
					chr_id = cd.chr_id, direction = cd.direction, mapq = cd.mapq, ref_bg = cd.ref_bg, align_score = cd.align_score, chain_score = cd.chain_score;
					cg = cigar_string(cd, J.cig);
				}
				if ((uint32_t)chr_id == 0xffffffffu) continue;           // primary_result->chrID == MAX_uint32_t
				int flag = (uint8_t)((k == 0 ? 0x40 : 0) + (direction == 0 ? 0x10 : 0) + (rr.has_mate ? 0 : 0x8));
				int isize = direction == 1 ? pr.cur_isize : -pr.cur_isize;
				std::string seq = rec.seq, qual = rec.qual;
				if (direction == 0) rev_seq(seq), rev_qual(qual);
				std::string tags;
				snprintf(b, sizeof b, "\tAS:i:%d", (int)align_score); tags += b;
				snprintf(b, sizeof b, "\tOS:i:%d\tOA:Z:%d,%d,%d,%d,%c;", (int)ori.align_score, ori.chr_id, (int)(ori.ref_bg >= 0x7fffffffu ? 1u : ori.ref_bg), (int)ori.read_bg, (int)ori.mapq,
				         rr.unmapped ? 'U' : 'M');
				tags += b;
				if (!is_ori) { snprintf(b, sizeof b, "\tCS:i:%d", (int)chain_score); tags += b; }
				const char *svs = psvr_index_sv_print_string(idx, rr.prim_sv_id);
				if (svs) tags += std::string("\tSV:Z:") + svs;
				const char *mvs = rr.has_mate ? psvr_index_sv_print_string(idx, rr.mate_sv_id) : nullptr;
				if (mvs) tags += std::string("\tMV:Z:") + mvs;
				if (rr.secondary >= 0) {
					const psvr_cand_t &sc = rr.cand[rr.secondary];
					const char *vid = psvr_index_sv_vcf_id(idx, sc.sv_id);
					snprintf(b, sizeof b, "\tXA:Z:%d,%d,%d,%d,%c,", sc.chr_id, (int)sc.ref_bg, (int)sc.read_bg, (int)sc.align_score, sc.direction == 1 ? 'F' : 'R');
					tags += b;
					tags += vid ? vid : "*";
					tags += ";";
				}
				tags += "\tRC:Z:" + rec.comment;
				emit_record(fo, dst, H, rec.name, flag, chr_id, ref_bg, mapq, cg, rr.has_mate != 0, rr.mate_chr_id, rr.mate_ref_bg, isize, seq, qual, tags);
			}
		}
		};
		// ---- second file (rr.cpp:776-797): pairs neither the original aligner nor the re-aligner placed well
		auto format_ori = [&](long long p0, long long p1, std::vector<uint8_t> &dst) {
		for (long long p = p0; p < p1; ++p) {
			const psvr_pair_result_t &pr = J.pres[p];
			if (!(pr.max_score <= par.min_filter_score && J.ori[2 * p].chr_id != -1 && J.ori[2 * p + 1].chr_id != -1)) continue;
			OriRecord orr[2];
			bool ok = parse_ori_record(J.recs[2 * p].comment, &orr[0]) && parse_ori_record(J.recs[2 * p + 1].comment, &orr[1]);
			if (!ok) continue;
			bool proper = pr.proper != 0;
			for (int k = 0; proper && k < 2; ++k) {
				const int mx = k == 0 ? pr.max1 : pr.max2;
				if (mx == -1) { proper = false; break; }
				if (mx == -2) { if (ori_has_clip(orr[k].cigar, 25)) proper = false; }
				else {                                               // bam_has_clip_or_unmapped_new (rr.cpp:735-743): sums the 'I' ops
					const psvr_cand_t &cd = J.res[2 * p + k].cand[mx];
					int tot = 0;
					for (uint32_t j = 0; j < cd.n_cigar; ++j) { uint32_t wv = J.cig[cd.cigar_off + j]; if ((wv & 0xf) == 1) tot += (int)(int16_t)(wv >> 4); }
					if (cd.n_cigar == 0 || tot >= 25) proper = false;
				}
			}
			if (proper) continue;
			for (int k = 0; k < 2; ++k) {
				const FqRec &rec = J.recs[2 * p + k];
				const psvr_ori_t &ori = J.ori[2 * p + k];
				std::string seq = rec.seq, qual = rec.qual;
				if (orr[k].flag & 0x10) rev_seq(seq), rev_qual(qual);
				std::string tags;
				if (!orr[k].tags.empty()) tags += "\t" + orr[k].tags;
				char b[64];
				snprintf(b, sizeof b, "\tMS:i:%d", pr.max_score);
				tags += b;
				const uint32_t ref_bg = ori.ref_bg >= 0x7fffffffu ? 1u : ori.ref_bg;
				emit_record(fo_ori, dst, H, rec.name, orr[k].flag, ori.chr_id, ref_bg + 1, orr[k].mapq, orr[k].cigar, true, orr[k].mate_chr, (uint32_t)orr[k].mate_pos, orr[k].isize, seq, qual, tags);
			}
		}
		};
		{
			const long long chunk = 4096, nchunk = (P + chunk - 1) / chunk;
			std::vector<std::vector<uint8_t>> mb(nchunk), ob(nchunk);
			std::atomic<long long> next(0);
			auto work = [&]() {
				for (long long ci = next++; ci < nchunk; ci = next++) {
					const long long p0 = ci * chunk, p1 = p0 + chunk < P ? p0 + chunk : P;
					format_main(p0, p1, mb[ci]), format_ori(p0, p1, ob[ci]);
				}
			};
			std::vector<std::thread> th;
			for (int t = 1; t < o.thread_n && t < nchunk; ++t) th.emplace_back(work);
			work();
			for (std::thread &t : th) t.join();
			t_format += walltime() - tw, tw = walltime();
			for (long long ci = 0; ci < nchunk; ++ci) fo.write_raw(mb[ci]), fo_ori.write_raw(ob[ci]);
			t_write += walltime() - tw;
		}
			set_state(J, 0);
		}
	});
	for (int slot = 0;; slot = (slot + 1) % 3) {
		Job &J = jobs[slot];
		wait_state(J, 1);
		if (J.last) { set_state(J, 2); break; }
		double tw = walltime();
		if (!eng) {
			fprintf(stderr, "Current used read status: READ_LEN=%d; ISIZE_MIN=%d; ISIZE_MID=%d; ISIZE_MAX=%d; filter_score_full_match=%d\n", par.normal_read_length, par.isize_min, 0,
			        par.isize_max, par.min_filter_score);
			if (psvr_engine_create(idx, &par, &eng)) { fprintf(stderr, "[panSVR-amd] %s\n", psvr_last_error()); abort(); }
		}
		const long long P = J.n_pairs(), R = 2 * P;
		J.res = (psvr_read_result_t *)J.res_buf.reserve((size_t)R * sizeof(psvr_read_result_t)), J.pres.resize(P);
		int rc = psvr_engine_upload(eng, P, J.bases.data(), (const int64_t *)J.base_off.data(), J.ori.data());
		if (!rc) rc = psvr_engine_run(eng, o.trace ? 1 : 0, nullptr);
		int64_t used = 0;
		if (!rc) { rc = psvr_engine_download(eng, nullptr, nullptr, nullptr, 0, &used); if (rc == PSVR_ERR_OVERFLOW) rc = 0; }
		J.cig = (uint32_t *)J.cig_buf.reserve((size_t)(used + 1) * 4);
		if (!rc) rc = psvr_engine_download(eng, J.res, J.pres.data(), J.cig, used + 1, &used);
		if (rc) { fprintf(stderr, "[panSVR-amd] engine error %d: %s\n", rc, psvr_last_error()); abort(); }
		t_engine += walltime() - tw;
		set_state(J, 2);
	}
	reader.join(), writer.join();
	if (fq != stdin) fclose(fq);
	if (sig_thread.joinable()) {
		sig_thread.join();
		if (sig_rc) { fprintf(stderr, "[panSVR-amd] the signal step failed\n"); abort(); }
	}
	if (!fo.close() || !fo_ori.close()) { fprintf(stderr, "fail to write output file\n"); abort(); }
	if (frec) fclose(frec);
	if (eng) psvr_engine_destroy(eng);
	psvr_index_destroy(idx);
	fprintf(stderr, "Classify CPU: %.3f sec\n", cputime() - cpu0);
	fprintf(stderr, "[panSVR-amd] wall: read+parse %.3f s, engine (upload+run+download) %.3f s, format %.3f s, write%s %.3f s\n", t_read, t_engine, t_format, o.sam ? "" : "+compress", t_write);
	return 0;
}
