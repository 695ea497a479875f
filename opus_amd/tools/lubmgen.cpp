// Deterministic LUBM-shaped N-Triples generator (host code, C ABI).
//
// The reference evaluates on LUBM data produced by the external UBA generator
// (reference README.md:62-78 only links it), which is not available offline.
// This tool emits data of the same shape: the univ-bench class / property
// vocabulary, the IRI scheme the reference's query constants use
// (queries/lubm100_1300_2300/Q1.txt:3-4, Q7.txt:7, Q8.txt:6) and one triple per
// line terminated by " ." so that the line format is exactly what
// Preprocessor.createInstanceGraph parses (Preprocessor.scala:119-122).
//
// Every university is generated from its own RNG stream (seed, university
// index), so any range [begin, end) can be produced independently and ranges
// concatenate to the same bytes as one big call: ranks of a multi-GPU job
// generate disjoint ranges without talking to each other.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

struct Rng {  // splitmix64: tiny, seedable per university
  uint64_t s;
  explicit Rng(uint64_t seed) : s(seed) {}
  uint64_t next() {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
  // uniform integer in [lo, hi]
  uint32_t range(uint32_t lo, uint32_t hi) { return lo + (uint32_t)(next() % (uint64_t)(hi - lo + 1)); }
};

struct Out {
  std::string buf;
  uint64_t triples = 0;
  void triple(const std::string& s, const std::string& p, const std::string& o) {
    buf += s; buf += ' '; buf += p; buf += ' '; buf += o; buf += " .\n";
    ++triples;
  }
};

const char* kRdfType = "<http://www.w3.org/1999/02/22-rdf-syntax-ns#type>";
const char* kOnto[2] = {
    "http://www.lehigh.edu/~zhp2/2004/0401/univ-bench.owl#",  // LUBM-100/1300/2300 query set
    "http://swat.cse.lehigh.edu/onto/univ-bench.owl#",        // LUBM-10240 query set
};

struct Vocab {
  std::string type, cls_univ, cls_dept, cls_full, cls_assoc, cls_asst, cls_lect, cls_ug, cls_gs, cls_ta, cls_ra,
      cls_course, cls_gcourse, cls_group, cls_pub;
  std::string name, email, phone, sub_org, works_for, teacher_of, ug_from, ms_from, phd_from, interest, head_of,
      member_of, takes, advisor, ta_of, pub_author;
  explicit Vocab(int onto) {
    std::string o = kOnto[onto];
    auto iri = [&](const char* local) { return "<" + o + local + ">"; };
    type = kRdfType;
    cls_univ = iri("University"); cls_dept = iri("Department"); cls_full = iri("FullProfessor");
    cls_assoc = iri("AssociateProfessor"); cls_asst = iri("AssistantProfessor"); cls_lect = iri("Lecturer");
    cls_ug = iri("UndergraduateStudent"); cls_gs = iri("GraduateStudent"); cls_ta = iri("TeachingAssistant");
    cls_ra = iri("ResearchAssistant"); cls_course = iri("Course"); cls_gcourse = iri("GraduateCourse");
    cls_group = iri("ResearchGroup"); cls_pub = iri("Publication");
    name = iri("name"); email = iri("emailAddress"); phone = iri("telephone"); sub_org = iri("subOrganizationOf");
    works_for = iri("worksFor"); teacher_of = iri("teacherOf"); ug_from = iri("undergraduateDegreeFrom");
    ms_from = iri("mastersDegreeFrom"); phd_from = iri("doctoralDegreeFrom"); interest = iri("researchInterest");
    head_of = iri("headOf"); member_of = iri("memberOf"); takes = iri("takesCourse"); advisor = iri("advisor");
    ta_of = iri("teachingAssistantOf"); pub_author = iri("publicationAuthor");
  }
};

std::string lit(const std::string& s) { return "\"" + s + "\""; }
std::string univ_iri(uint32_t u) { return "<http://www.University" + std::to_string(u) + ".edu>"; }

struct Faculty { std::string iri; int rank; std::vector<int> gcourses; };

void person_literals(Out& out, const Vocab& v, Rng& rng, const std::string& iri, const std::string& local,
                     const std::string& host) {
  out.triple(iri, v.name, lit(local));
  out.triple(iri, v.email, lit(local + "@" + host));
  char tel[32];
  snprintf(tel, sizeof tel, "%03u-%03u-%04u", rng.range(100, 999), rng.range(100, 999), rng.range(0, 9999));
  out.triple(iri, v.phone, lit(tel));
}

void gen_university(Out& out, const Vocab& v, uint64_t seed, uint32_t u, uint32_t degree_pool) {
  Rng rng(seed * 0x100000001B3ull + 0xC0FFEEull + (uint64_t)u * 0x9E3779B97F4A7C15ull);
  const std::string U = univ_iri(u);
  out.triple(U, v.type, v.cls_univ);
  out.triple(U, v.name, lit("University" + std::to_string(u)));
  const uint32_t ndept = rng.range(15, 25);
  for (uint32_t d = 0; d < ndept; ++d) {
    const std::string host = "Department" + std::to_string(d) + ".University" + std::to_string(u) + ".edu";
    const std::string D = "<http://www." + host + ">";
    const std::string base = "<http://www." + host + "/";
    out.triple(D, v.type, v.cls_dept);
    out.triple(D, v.name, lit("Department" + std::to_string(d)));
    out.triple(D, v.sub_org, U);

    const char* rank_name[4] = {"FullProfessor", "AssociateProfessor", "AssistantProfessor", "Lecturer"};
    const std::string* rank_cls[4] = {&v.cls_full, &v.cls_assoc, &v.cls_asst, &v.cls_lect};
    const uint32_t nrank[4] = {rng.range(7, 10), rng.range(10, 14), rng.range(8, 11), rng.range(5, 7)};
    std::vector<Faculty> fac;
    int ncourse = 0, ngcourse = 0;
    for (int r = 0; r < 4; ++r) {
      for (uint32_t i = 0; i < nrank[r]; ++i) {
        const std::string local = std::string(rank_name[r]) + std::to_string(i);
        Faculty f; f.iri = base + local + ">"; f.rank = r;
        out.triple(f.iri, v.type, *rank_cls[r]);
        person_literals(out, v, rng, f.iri, local, host);
        out.triple(f.iri, v.works_for, D);
        const uint32_t nc = rng.range(1, 2), ngc = rng.range(1, 2);
        for (uint32_t c = 0; c < nc; ++c) {
          const std::string C = base + "Course" + std::to_string(ncourse) + ">";
          out.triple(f.iri, v.teacher_of, C);
          out.triple(C, v.type, v.cls_course);
          out.triple(C, v.name, lit("Course" + std::to_string(ncourse)));
          ++ncourse;
        }
        for (uint32_t c = 0; c < ngc; ++c) {
          const std::string C = base + "GraduateCourse" + std::to_string(ngcourse) + ">";
          out.triple(f.iri, v.teacher_of, C);
          out.triple(C, v.type, v.cls_gcourse);
          out.triple(C, v.name, lit("GraduateCourse" + std::to_string(ngcourse)));
          f.gcourses.push_back(ngcourse);
          ++ngcourse;
        }
        out.triple(f.iri, v.ug_from, univ_iri(rng.range(0, degree_pool - 1)));
        out.triple(f.iri, v.ms_from, univ_iri(rng.range(0, degree_pool - 1)));
        out.triple(f.iri, v.phd_from, univ_iri(rng.range(0, degree_pool - 1)));
        if (r < 3) out.triple(f.iri, v.interest, lit("Research" + std::to_string(rng.range(0, 29))));
        if (r == 0 && i == 0) out.triple(f.iri, v.head_of, D);
        fac.push_back(std::move(f));
      }
    }
    const uint32_t nfac = (uint32_t)fac.size();
    const uint32_t nprof = nrank[0] + nrank[1] + nrank[2];

    const uint32_t nug = nfac * rng.range(8, 14);
    for (uint32_t i = 0; i < nug; ++i) {
      const std::string local = "UndergraduateStudent" + std::to_string(i);
      const std::string S = base + local + ">";
      out.triple(S, v.type, v.cls_ug);
      person_literals(out, v, rng, S, local, host);
      out.triple(S, v.member_of, D);
      const uint32_t nt = rng.range(2, 4);
      uint32_t first = rng.range(0, (uint32_t)ncourse - 1);
      for (uint32_t c = 0; c < nt; ++c)  // consecutive course indices: distinct without a set
        out.triple(S, v.takes, base + "Course" + std::to_string((first + c) % (uint32_t)ncourse) + ">");
      if (rng.range(0, 4) == 0) out.triple(S, v.advisor, fac[rng.range(0, nprof - 1)].iri);
    }

    const uint32_t ngs = nfac * rng.range(3, 4);
    std::vector<std::vector<std::string>> advisees(nprof);
    for (uint32_t i = 0; i < ngs; ++i) {
      const std::string local = "GraduateStudent" + std::to_string(i);
      const std::string S = base + local + ">";
      out.triple(S, v.type, v.cls_gs);
      const uint32_t extra = rng.range(0, 19);
      if (extra < 4) out.triple(S, v.type, v.cls_ta);
      else if (extra < 9) out.triple(S, v.type, v.cls_ra);
      person_literals(out, v, rng, S, local, host);
      out.triple(S, v.member_of, D);
      out.triple(S, v.ug_from, univ_iri(rng.range(0, degree_pool - 1)));
      const uint32_t adv = rng.range(0, nprof - 1);
      out.triple(S, v.advisor, fac[adv].iri);
      advisees[adv].push_back(S);
      const uint32_t nt = rng.range(1, 3);
      // Student 0 always takes GraduateCourse0 (query constant of Q1/Q10); one in
      // eight students starts at a course their advisor teaches (Q9's triangle).
      uint32_t first = (i == 0) ? 0u
                       : (rng.range(0, 7) == 0 ? (uint32_t)fac[adv].gcourses[0] : rng.range(0, (uint32_t)ngcourse - 1));
      for (uint32_t c = 0; c < nt; ++c)
        out.triple(S, v.takes, base + "GraduateCourse" + std::to_string((first + c) % (uint32_t)ngcourse) + ">");
      if (extra < 4)
        out.triple(S, v.ta_of, base + "Course" + std::to_string(rng.range(0, (uint32_t)ncourse - 1)) + ">");
    }

    const uint32_t ngroup = rng.range(10, 20);
    for (uint32_t i = 0; i < ngroup; ++i) {
      const std::string G = base + "ResearchGroup" + std::to_string(i) + ">";
      out.triple(G, v.type, v.cls_group);
      out.triple(G, v.sub_org, D);
    }

    const uint32_t pub_lo[4] = {15, 10, 5, 0}, pub_hi[4] = {20, 18, 10, 5};
    for (uint32_t fi = 0; fi < nfac; ++fi) {
      const Faculty& f = fac[fi];
      const uint32_t np = rng.range(pub_lo[f.rank], pub_hi[f.rank]);
      const std::string stem = f.iri.substr(0, f.iri.size() - 1) + "/Publication";
      for (uint32_t i = 0; i < np; ++i) {
        const std::string P = stem + std::to_string(i) + ">";
        out.triple(P, v.type, v.cls_pub);
        out.triple(P, v.name, lit("Publication" + std::to_string(i)));
        out.triple(P, v.pub_author, f.iri);
        if (fi < nprof && !advisees[fi].empty() && rng.range(0, 2) == 0)
          out.triple(P, v.pub_author, advisees[fi][rng.range(0, (uint32_t)advisees[fi].size() - 1)]);
      }
    }
  }
}

}  // namespace

extern "C" {

// Generates universities [univ_begin, univ_end). ontology: 0 = the namespace of
// queries/lubm100_1300_2300, 1 = the namespace of queries/lubm10240. degree_pool
// is the number of universities degrees are drawn from (UBA uses 1000). Returns
// a malloc'ed buffer (free with lubm_free); *n_bytes and *n_triples are set.
char* lubm_generate(uint32_t univ_begin, uint32_t univ_end, uint64_t seed, int ontology, uint32_t degree_pool,
                    uint64_t* n_bytes, uint64_t* n_triples) {
  if (ontology < 0 || ontology > 1 || degree_pool == 0 || univ_end < univ_begin) return nullptr;
  Vocab v(ontology);
  Out out;
  const size_t guess = (size_t)(univ_end - univ_begin) * 26000000ull;  // ~24 MB of text per university
  out.buf.reserve(guess < ((size_t)3 << 30) ? guess : ((size_t)3 << 30));
  for (uint32_t u = univ_begin; u < univ_end; ++u) gen_university(out, v, seed, u, degree_pool);
  char* p = (char*)malloc(out.buf.size() ? out.buf.size() : 1);
  if (!p) return nullptr;
  memcpy(p, out.buf.data(), out.buf.size());
  if (n_bytes) *n_bytes = out.buf.size();
  if (n_triples) *n_triples = out.triples;
  return p;
}

void lubm_free(char* p) { free(p); }

}  // extern "C"

#ifdef LUBMGEN_MAIN
// lubmgen <univ_begin> <univ_end> <seed> <ontology 0|1> <degree_pool> > out.nt
int main(int argc, char** argv) {
  if (argc != 6) { fprintf(stderr, "usage: lubmgen begin end seed ontology degree_pool\n"); return 2; }
  uint64_t nb = 0, nt = 0;
  char* p = lubm_generate((uint32_t)atoi(argv[1]), (uint32_t)atoi(argv[2]), strtoull(argv[3], 0, 10), atoi(argv[4]),
                          (uint32_t)atoi(argv[5]), &nb, &nt);
  if (!p) return 1;
  fwrite(p, 1, nb, stdout);
  fprintf(stderr, "%llu triples, %llu bytes\n", (unsigned long long)nt, (unsigned long long)nb);
  lubm_free(p);
  return 0;
}
#endif
