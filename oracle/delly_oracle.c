/*
 * TEST INFRASTRUCTURE ONLY -- see delly_oracle.h.  CPU restatement (plain C)
 * of the reference's split-read refinement path; each function cites the
 * reference file:line it follows (paths relative to /root/reference).
 * Parity pinned against oracle/_ref (the reference's own headers).
 */
#define _GNU_SOURCE
#include "delly_oracle.h"

#include <ctype.h>
#include <pthread.h>
#include <stdatomic.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------------ */
/* small helpers                                                             */

static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int imin(int a, int b) { return a < b ? a : b; }

typedef struct {
  char* d; /* row-major rows x cols */
  int rows, cols;
} amat;

static amat amat_new(int rows, int cols) {
  amat a;
  a.rows = rows;
  a.cols = cols;
  a.d = (char*)calloc((size_t)rows * (size_t)(cols > 0 ? cols : 1) + 1, 1);
  return a;
}
static void amat_free(amat* a) {
  free(a->d);
  a->d = NULL;
}
#define AT(a, i, j) ((a).d[(size_t)(i) * (size_t)(a).cols + (size_t)(j)])

/* src/util.h:549-563 reverseComplement: upper-cased reverse, complemented; a
 * letter outside ACGTN leaves the ORIGINAL (un-reversed) byte at index i. */
void dor_reverse_complement(char* s, int n) {
  char* up = (char*)malloc((size_t)n + 1);
  for (int i = 0; i < n; ++i) up[i] = (char)toupper((unsigned char)s[n - 1 - i]);
  for (int i = 0; i < n; ++i) {
    switch (up[i]) {
      case 'A': s[i] = 'T'; break;
      case 'C': s[i] = 'G'; break;
      case 'G': s[i] = 'C'; break;
      case 'T': s[i] = 'A'; break;
      case 'N': s[i] = 'N'; break;
      default: break;
    }
  }
  free(up);
}

/* ------------------------------------------------------------------------ */
/* K1: lcs  src/msa.h:10-30                                                   */

int dor_lcs(const char* s1, int m, const char* s2, int n) {
  int32_t* onecol = (int32_t*)calloc((size_t)n + 1, sizeof(int32_t));
  int32_t prevdiag = 0;
  for (int i = 0; i <= m; ++i) {
    for (int j = 0; j <= n; ++j) {
      if (i == 0 || j == 0) {
        onecol[j] = 0;
        prevdiag = 0;
      } else {
        int32_t prevprevdiag = prevdiag;
        prevdiag = onecol[j];
        if (s1[i - 1] == s2[j - 1]) onecol[j] = prevprevdiag + 1;
        else onecol[j] = (onecol[j] > onecol[j - 1]) ? onecol[j] : onecol[j - 1];
      }
    }
  }
  int r = onecol[n];
  free(onecol);
  return r;
}

/* ------------------------------------------------------------------------ */
/* K4: longestHomology  src/needle.h:13-42 (band k=|threshold|)               */

int dor_longest_homology(const char* s1, int m, const char* s2, int n, int thr) {
  size_t W = (size_t)n + 3;
  int32_t* mat = (int32_t*)calloc(((size_t)m + 3) * W, sizeof(int32_t));
  int k = abs(thr);
  mat[0] = 0;
  for (int col = 1; col <= k; ++col) mat[col] = mat[col - 1] - 1;
  for (int row = 1; row <= k; ++row) mat[(size_t)row * W] = mat[(size_t)(row - 1) * W] - 1;
  int ret = 0;
  int done = 0;
  for (int row = 1; row <= m && !done; ++row) {
    int bestCol = thr - 1;
    for (int h = -k; h <= k; ++h) {
      int col = row + h;
      if (col >= 1 && col <= n) {
        int32_t* c = &mat[(size_t)row * W + col];
        *c = mat[(size_t)(row - 1) * W + col - 1] + (s1[row - 1] == s2[col - 1] ? 0 : -1);
        if ((row - 1 - col >= -k) && (row - 1 - col <= k)) *c = imax(*c, mat[(size_t)(row - 1) * W + col] - 1);
        if ((row - col + 1 >= -k) && (row - col + 1 <= k)) *c = imax(*c, mat[(size_t)row * W + col - 1] - 1);
        if (*c > bestCol) bestCol = *c;
      }
    }
    if (bestCol < thr) {
      ret = row - 1;
      done = 1;
    }
  }
  free(mat);
  return ret;
}

/* ------------------------------------------------------------------------ */
/* K3: longNeedle  src/needle.h:45-222 with AlignConfig<true,false> and       */
/* DnaScore(1,-1,-1,-1) as fixed by src/split.h:543-555.                      */

static inline int hgap(int row, int m) { return (row == 0 || row == m) ? 0 : -1; } /* align.h:67-73 */

static void fill_needle(const char* a, int m, const char* b, int n, int32_t* mat) {
  size_t W = (size_t)n + 1;
  mat[0] = 0;
  for (int col = 1; col <= n; ++col) mat[col] = mat[col - 1] + hgap(0, m);
  for (int row = 1; row <= m; ++row) mat[(size_t)row * W] = mat[(size_t)(row - 1) * W] + (-1);
  for (int row = 1; row <= m; ++row) {
    int hg = hgap(row, m);
    const int32_t* up = &mat[(size_t)(row - 1) * W];
    int32_t* cur = &mat[(size_t)row * W];
    for (int col = 1; col <= n; ++col) {
      int d = up[col - 1] + (a[row - 1] == b[col - 1] ? 1 : -1);
      int v = up[col] + (-1);
      int h = cur[col - 1] + hg;
      cur[col] = imax(imax(d, v), h);
    }
  }
}

/* traceback of src/needle.h:159-171 / :180-192; returns trace length, ops in
 * push order (from the end of the path backwards) */
static int trace_needle(const int32_t* mat, int m, int n, int rr, int cc, char* trace) {
  size_t W = (size_t)n + 1;
  int t = 0;
  while (rr > 0 || cc > 0) {
    if (rr > 0 && mat[(size_t)rr * W + cc] == mat[(size_t)(rr - 1) * W + cc] + (-1)) {
      --rr;
      trace[t++] = 'v';
    } else if (cc > 0 && mat[(size_t)rr * W + cc] == mat[(size_t)rr * W + cc - 1] + hgap(rr, m)) {
      --cc;
      trace[t++] = 'h';
    } else {
      --rr;
      --cc;
      trace[t++] = 's';
    }
  }
  return t;
}

/* src/align.h:173-200 _createAlignment for two strings */
static amat make_alignment(const char* trace, int tlen, const char* s1, const char* s2) {
  amat al = amat_new(2, tlen);
  int row = 0, col = 0, ai = 0;
  for (int t = tlen - 1; t >= 0; --t, ++ai) {
    if (trace[t] == 's') {
      AT(al, 0, ai) = s1[row++];
      AT(al, 1, ai) = s2[col++];
    } else if (trace[t] == 'h') {
      AT(al, 0, ai) = '-';
      AT(al, 1, ai) = s2[col++];
    } else {
      AT(al, 0, ai) = s1[row++];
      AT(al, 1, ai) = '-';
    }
  }
  return al;
}

/* returns 1 found, 0 not found; *out receives the 2 x len alignment */
static int long_needle_core(const char* s1, int m, const char* s2, int n, amat* out, int* diag) {
  size_t W = (size_t)n + 1;
  size_t cells = ((size_t)m + 1) * W;
  int32_t* mat = (int32_t*)malloc(cells * sizeof(int32_t));
  int32_t* rev = (int32_t*)malloc(cells * sizeof(int32_t));
  char* r1 = (char*)malloc((size_t)m + 1);
  char* r2 = (char*)malloc((size_t)n + 1);
  memcpy(r1, s1, (size_t)m);
  memcpy(r2, s2, (size_t)n);
  dor_reverse_complement(r1, m);
  dor_reverse_complement(r2, n);
  fill_needle(s1, m, s2, n, mat);
  fill_needle(r1, m, r2, n, rev);
  int found = 0;
  if (diag) {
    diag[0] = mat[cells - 1];
    diag[1] = diag[2] = diag[3] = diag[4] = -1;
  }
  out->d = NULL;
  out->rows = 2;
  out->cols = 0;
  if (mat[cells - 1] == rev[cells - 1]) {
    /* best join, needle.h:87-123 */
    int32_t* bm = (int32_t*)malloc(cells * sizeof(int32_t));
    int32_t* br = (int32_t*)malloc(cells * sizeof(int32_t));
    for (int row = 0; row <= m; ++row) {
      bm[(size_t)row * W] = mat[(size_t)row * W];
      br[(size_t)row * W] = rev[(size_t)row * W];
      for (int col = 1; col <= n; ++col) {
        size_t i = (size_t)row * W + col;
        bm[i] = (mat[i] > bm[i - 1]) ? mat[i] : bm[i - 1];
        br[i] = (rev[i] > br[i - 1]) ? rev[i] : br[i - 1];
      }
    }
    int best = mat[cells - 1];
    int consLeft = 0, refLeft = 0;
    for (int row = 0; row <= m; ++row)
      for (int col = 0; col <= n; ++col) {
        int v = bm[(size_t)row * W + col] + br[(size_t)(m - row) * W + (n - col)];
        if (v > best) {
          best = v;
          consLeft = row;
          refLeft = col;
        }
      }
    int consRight = m - consLeft;
    int refRight = 0;
    for (int right = 0; right <= n - refLeft; ++right)
      if (mat[(size_t)consLeft * W + refLeft] + rev[(size_t)consRight * W + right] == best) refRight = right;
    free(bm);
    free(br);
    if (diag) {
      diag[1] = best;
      diag[2] = consLeft;
      diag[3] = refLeft;
      diag[4] = refRight;
    }
    if (best != mat[cells - 1]) {
      char* tr = (char*)malloc((size_t)m + n + 2);
      int tl = trace_needle(mat, m, n, consLeft, refLeft, tr);
      amat fwd = make_alignment(tr, tl, s1, s2);
      int rl = trace_needle(rev, m, n, consRight, refRight, tr);
      amat rvs = make_alignment(tr, rl, r1, r2);
      free(tr);
      /* concat, needle.h:196-219 */
      int gapref = (n - refRight) - refLeft;
      int alilen = fwd.cols + rvs.cols + gapref;
      amat al = amat_new(2, alilen);
      int jEnd = rvs.cols;
      for (int i = 0; i < 2; ++i) {
        int ac = 0;
        for (; ac < fwd.cols; ++ac) AT(al, i, ac) = AT(fwd, i, ac);
        for (int j = refLeft; j < n - refRight; ++j, ++ac) AT(al, i, ac) = (i == 0) ? '-' : s2[j];
        for (int j = 0; j < rvs.cols; ++j, ++ac) {
          switch (AT(rvs, i, jEnd - j - 1)) {
            case 'A': AT(al, i, ac) = 'T'; break;
            case 'C': AT(al, i, ac) = 'G'; break;
            case 'G': AT(al, i, ac) = 'C'; break;
            case 'T': AT(al, i, ac) = 'A'; break;
            case 'N': AT(al, i, ac) = 'N'; break;
            case '-': AT(al, i, ac) = '-'; break;
            default: break; /* stays '\0' (value-initialised multi_array) */
          }
        }
      }
      amat_free(&fwd);
      amat_free(&rvs);
      *out = al;
      found = 1;
    }
  }
  free(mat);
  free(rev);
  free(r1);
  free(r2);
  return found;
}

int dor_long_needle(const char* s1, int m, const char* s2, int n, char* rows, int cap, int* len,
                    int* diag) {
  amat al;
  int found = long_needle_core(s1, m, s2, n, &al, diag);
  *len = 0;
  if (!found) return 0;
  *len = al.cols;
  int rc = 1;
  if (al.cols > cap) rc = -1;
  else
    for (int j = 0; j < al.cols; ++j) {
      rows[j] = AT(al, 0, j);
      rows[(size_t)cap + j] = AT(al, 1, j);
    }
  amat_free(&al);
  return rc;
}

/* ------------------------------------------------------------------------ */
/* K2: profile Gotoh  src/gotoh.h:71-174, src/align.h:89-171,202-229          */
/* AlignConfig<true,true> (src/msa.h:106): both end gaps free.                */

/* src/align.h:131-171 _createProfile: 6 x cols float (A,C,G,T,N,-) */
static float* create_profile(const amat* a) {
  int R = a->rows, C = a->cols;
  float* p = (float*)calloc((size_t)6 * (size_t)(C > 0 ? C : 1), sizeof(float));
  int32_t* first = (int32_t*)malloc(sizeof(int32_t) * (size_t)(R > 0 ? R : 1));
  int32_t* last = (int32_t*)malloc(sizeof(int32_t) * (size_t)(R > 0 ? R : 1));
  for (int i = 0; i < R; ++i) {
    first[i] = -1;
    last[i] = C;
    for (int j = 0; j < C; ++j) {
      if (first[i] == -1) {
        if (AT(*a, i, j) != '-') first[i] = j;
      }
      if (first[i] != -1) {
        if (AT(*a, i, j) != '-') last[i] = j;
      }
    }
  }
  for (int j = 0; j < C; ++j) {
    int sum = 0;
    for (int i = 0; i < R; ++i) {
      if (first[i] <= j && j <= last[i]) {
        ++sum;
        char ch = AT(*a, i, j);
        if (ch == 'A' || ch == 'a') p[0 * (size_t)C + j] += 1;
        else if (ch == 'C' || ch == 'c') p[1 * (size_t)C + j] += 1;
        else if (ch == 'G' || ch == 'g') p[2 * (size_t)C + j] += 1;
        else if (ch == 'T' || ch == 't') p[3 * (size_t)C + j] += 1;
        else if (ch == 'N' || ch == 'n') p[4 * (size_t)C + j] += 1;
        else if (ch == '-') p[5 * (size_t)C + j] += 1;
        else --sum;
      }
    }
    for (int k = 0; k < 6; ++k) p[k * (size_t)C + j] /= sum; /* float / int -> float division */
  }
  free(first);
  free(last);
  return p;
}

static inline int gap_free(int i, int iend, int cost) { return (i == 0 || i == iend) ? 0 : cost; }

/* returns score; *out = (r1+r2) x len alignment */
static int gotoh_core(const dellyhip_params* sc, const amat* a1, const amat* a2, amat* out) {
  int m = a1->cols, n = a2->cols;
  const int inf = 1000000; /* DnaScore::inf, align.h:21 */
  size_t mf = (size_t)n + 1;
  size_t cells = ((size_t)m + 1) * mf;
  int32_t* s = (int32_t*)calloc(mf, sizeof(int32_t));
  int32_t* v = (int32_t*)calloc(mf, sizeof(int32_t));
  uint8_t* bits = (uint8_t*)calloc(cells, 1); /* bit0..3 = bit1..bit4 of gotoh.h:88-91 */
  int32_t newhoz = 0, prevsub = 0;
  int single = (a1->rows == 1 && a2->rows == 1);
  float *p1 = NULL, *p2 = NULL;
  if (!single) {
    p1 = create_profile(a1);
    p2 = create_profile(a2);
  }
  int go = sc->gap_open, ge = sc->gap_extend;
  for (int row = 0; row <= m; ++row) {
    for (int col = 0; col <= n; ++col) {
      if (row == 0 && col == 0) {
        s[0] = 0;
        v[0] = -inf;
        newhoz = -inf;
        bits[0] |= 1 | 2;
      } else if (row == 0) {
        v[col] = -inf;
        s[col] = gap_free(0, m, go + col * ge);
        newhoz = gap_free(0, m, go + col * ge);
        bits[col] |= 4;
      } else if (col == 0) {
        newhoz = -inf;
        s[0] = gap_free(0, n, go + row * ge);
        if (row - 1 == 0) prevsub = 0;
        else prevsub = gap_free(0, n, go + (row - 1) * ge);
        v[0] = gap_free(0, n, go + row * ge);
        bits[(size_t)row * mf] |= 8;
      } else {
        int32_t prevhoz = newhoz;
        int32_t prevver = v[col];
        int32_t prevprevsub = prevsub;
        prevsub = s[col];
        newhoz = imax(s[col - 1] + gap_free(row, m, go + ge), prevhoz + gap_free(row, m, ge));
        v[col] = imax(prevsub + gap_free(col, n, go + ge), prevver + gap_free(col, n, ge));
        int sco;
        if (single) sco = (AT(*a1, 0, row - 1) == AT(*a2, 0, col - 1)) ? sc->match : sc->mismatch;
        else {
          /* align.h:105-109: float accumulation in this exact order */
          float score = 0;
          for (int k1 = 0; k1 < 5; ++k1)
            for (int k2 = 0; k2 < 5; ++k2)
              score += p1[(size_t)k1 * (size_t)m + (row - 1)] * p2[(size_t)k2 * (size_t)n + (col - 1)] *
                       ((k1 == k2) ? sc->match : sc->mismatch);
          sco = (int)score;
        }
        s[col] = imax(imax(prevprevsub + sco, newhoz), v[col]);
        uint8_t b = 0;
        if (s[col] == newhoz) b |= 4;
        else if (s[col] == v[col]) b |= 8;
        if (newhoz != prevhoz + gap_free(row, m, ge)) b |= 1;
        if (v[col] != prevver + gap_free(col, n, ge)) b |= 2;
        bits[(size_t)row * mf + col] |= b;
      }
    }
  }
  int score = s[n];
  /* traceback, gotoh.h:143-167 */
  char* btr = (char*)malloc((size_t)m + n + 2);
  int tl = 0;
  int row = m, col = n;
  char last = 's';
  while (row > 0 || col > 0) {
    uint8_t b = bits[(size_t)row * mf + col];
    if (last == 's') {
      if (b & 4) last = 'h';
      else if (b & 8) last = 'v';
      else {
        --row;
        --col;
        btr[tl++] = 's';
      }
    } else if (last == 'h') {
      if (b & 1) last = 's';
      --col;
      btr[tl++] = 'h';
    } else {
      if (b & 2) last = 's';
      --row;
      btr[tl++] = 'v';
    }
  }
  /* _createAlignment, align.h:202-229 */
  int numN = a1->rows, numM = a2->rows;
  amat al = amat_new(numN + numM, tl);
  int r = 0, c = 0, ai = 0;
  for (int t = tl - 1; t >= 0; --t, ++ai) {
    if (btr[t] == 's') {
      for (int i = 0; i < numN; ++i) AT(al, i, ai) = AT(*a1, i, r);
      for (int i = 0; i < numM; ++i) AT(al, numN + i, ai) = AT(*a2, i, c);
      ++r;
      ++c;
    } else if (btr[t] == 'h') {
      for (int i = 0; i < numN; ++i) AT(al, i, ai) = '-';
      for (int i = 0; i < numM; ++i) AT(al, numN + i, ai) = AT(*a2, i, c);
      ++c;
    } else {
      for (int i = 0; i < numN; ++i) AT(al, i, ai) = AT(*a1, i, r);
      for (int i = 0; i < numM; ++i) AT(al, numN + i, ai) = '-';
      ++r;
    }
  }
  free(btr);
  free(bits);
  free(s);
  free(v);
  free(p1);
  free(p2);
  *out = al;
  return score;
}

int dor_gotoh(const dellyhip_params* p, const char* a1, int r1, int m, const char* a2, int r2, int n,
              char* out, int cap, int* len) {
  amat A1 = amat_new(r1, m), A2 = amat_new(r2, n), A;
  memcpy(A1.d, a1, (size_t)r1 * m);
  memcpy(A2.d, a2, (size_t)r2 * n);
  int score = gotoh_core(p, &A1, &A2, &A);
  *len = A.cols;
  if (A.cols <= cap)
    for (int i = 0; i < A.rows; ++i) memcpy(out + (size_t)i * cap, A.d + (size_t)i * A.cols, (size_t)A.cols);
  amat_free(&A1);
  amat_free(&A2);
  amat_free(&A);
  return score;
}

/* ------------------------------------------------------------------------ */
/* K7: consensus  src/msa.h:111-173                                           */

static int consensus_core(const dellyhip_params* p, const amat* al, char* cs /* cap >= cols */) {
  int R = al->rows, C = al->cols;
  uint8_t* fl = (uint8_t*)calloc((size_t)(R > 0 ? R : 1) * (size_t)(C > 0 ? C : 1), 1);
  int* cov = (int*)calloc((size_t)(C > 0 ? C : 1), sizeof(int));
  for (int i = 0; i < R; ++i) {
    int start = 0, end = -1;
    for (int j = 0; j < C; ++j) {
      if (AT(*al, i, j) != '-') end = j;
      else if (end == -1) start = j + 1;
    }
    for (int j = start; j <= end; ++j) {
      ++cov[j];
      fl[(size_t)i * C + j] = 1;
    }
  }
  int thr = imax(2, imin(p->min_clique_size, R));
  int L = 0;
  for (int j = 0; j < C; ++j) {
    int maxIdx = 4;
    if (cov[j] >= thr) {
      int count[5] = {0, 0, 0, 0, 0};
      for (int i = 0; i < R; ++i)
        if (fl[(size_t)i * C + j]) {
          char ch = AT(*al, i, j);
          if (ch == 'A' || ch == 'a') ++count[0];
          else if (ch == 'C' || ch == 'c') ++count[1];
          else if (ch == 'G' || ch == 'g') ++count[2];
          else if (ch == 'T' || ch == 't') ++count[3];
          else ++count[4];
        }
      maxIdx = 0;
      int maxCount = count[0];
      for (int i = 1; i < 5; ++i)
        if (count[i] > maxCount) {
          maxCount = count[i];
          maxIdx = i;
        }
    }
    if (maxIdx < 4) cs[L++] = "ACGT"[maxIdx];
  }
  free(fl);
  free(cov);
  return L;
}

int dor_consensus(const dellyhip_params* p, const char* a, int r, int m, char* cs, int cap) {
  amat A = amat_new(r, m);
  memcpy(A.d, a, (size_t)r * m);
  char* tmp = (char*)malloc((size_t)m + 1);
  int L = consensus_core(p, &A, tmp);
  if (L <= cap) memcpy(cs, tmp, (size_t)L);
  free(tmp);
  amat_free(&A);
  return L;
}

/* ------------------------------------------------------------------------ */
/* guide tree: distanceMatrix + upgma  src/msa.h:32-89                        */

static int guide_tree(int num, const char* blob, const uint64_t* off, int* d /* (2num+1)^2 */,
                      int* p /* (2num+1) x 3 */, int* dcopy /* distance matrix before UPGMA, or NULL */) {
  int D = 2 * num + 1;
  for (int i = 0; i < D; ++i)
    for (int j = 0; j < D; ++j) d[i * D + j] = (j > i) ? -1 : 0; /* msa.h:192-195 (rest value-init 0) */
  for (int i = 0; i < num; ++i)
    for (int j = i + 1; j < num; ++j) {
      int li = (int)(off[i + 1] - off[i]), lj = (int)(off[j + 1] - off[j]);
      int l = dor_lcs(blob + off[i], li, blob + off[j], lj);
      uint64_t mn = (uint64_t)(li < lj ? li : lj);
      d[i * D + j] = (int)(((uint64_t)(int64_t)(l * 100)) / mn); /* msa.h:41: int*100 / size_t */
    }
  if (dcopy) memcpy(dcopy, d, sizeof(int) * (size_t)D * D);
  for (int i = 0; i < D; ++i) p[i * 3 + 0] = p[i * 3 + 1] = p[i * 3 + 2] = -1;
  int nn = num;
  for (; nn < 2 * num + 1; ++nn) {
    /* closestPair msa.h:46-60 */
    int dMax = -1, dI = 0, dJ = 0;
    for (int i = 0; i < nn; ++i)
      for (int j = i + 1; j < nn; ++j)
        if (d[i * D + j] > dMax) {
          dMax = d[i * D + j];
          dI = i;
          dJ = j;
        }
    if (dMax == -1) break;
    p[dI * 3] = nn;
    p[dJ * 3] = nn;
    p[nn * 3 + 1] = dI;
    p[nn * 3 + 2] = dJ;
    /* updateDistanceMatrix msa.h:62-72 */
    for (int i = 0; i < nn; ++i)
      if (p[i * 3] == -1)
        d[i * D + nn] = (((dI < i) ? d[dI * D + i] : d[i * D + dI]) + ((dJ < i) ? d[dJ * D + i] : d[i * D + dJ])) / 2;
    for (int i = 0; i < dI; ++i) d[i * D + dI] = -1;
    for (int i = dI + 1; i < nn + 1; ++i) d[dI * D + i] = -1;
    for (int i = 0; i < dJ; ++i) d[i * D + dJ] = -1;
    for (int i = dJ + 1; i < nn + 1; ++i) d[dJ * D + i] = -1;
  }
  return (nn > 0) ? (nn - 1) : 0;
}

int dor_guide_tree(int n_reads, const char* blob, const uint64_t* off, int* dflat, int* pflat) {
  int D = 2 * n_reads + 1;
  int* d = (int*)malloc(sizeof(int) * (size_t)D * D);
  int root = guide_tree(n_reads, blob, off, d, pflat, dflat);
  free(d);
  return root;
}

/* palign  src/msa.h:91-109 */
static amat palign(const dellyhip_params* p, const char* blob, const uint64_t* off, const int* ph,
                   int root) {
  if (ph[root * 3 + 1] == -1 && ph[root * 3 + 2] == -1) {
    int L = (int)(off[root + 1] - off[root]);
    amat a = amat_new(1, L);
    memcpy(a.d, blob + off[root], (size_t)L);
    return a;
  }
  amat a1 = palign(p, blob, off, ph, ph[root * 3 + 1]);
  amat a2 = palign(p, blob, off, ph, ph[root * 3 + 2]);
  amat out;
  gotoh_core(p, &a1, &a2, &out);
  amat_free(&a1);
  amat_free(&a2);
  return out;
}

/* msa  src/msa.h:185-239 */
static int msa_core(const dellyhip_params* p, int n_reads, const char* blob, const uint64_t* off,
                    char** cs, int* cs_len) {
  int D = 2 * n_reads + 1;
  int* d = (int*)malloc(sizeof(int) * (size_t)D * D);
  int* ph = (int*)malloc(sizeof(int) * (size_t)D * 3);
  int root = guide_tree(n_reads, blob, off, d, ph, NULL);
  amat al = palign(p, blob, off, ph, root);
  *cs = (char*)malloc((size_t)al.cols + 1);
  *cs_len = consensus_core(p, &al, *cs);
  int rows = al.rows;
  amat_free(&al);
  free(d);
  free(ph);
  return rows;
}

int dor_msa(const dellyhip_params* p, int n_reads, const char* blob, const uint64_t* off, char* cs,
            int cap, int* cs_len) {
  char* tmp;
  int rows = msa_core(p, n_reads, blob, off, &tmp, cs_len);
  if (*cs_len <= cap) memcpy(cs, tmp, (size_t)*cs_len);
  free(tmp);
  return rows;
}

/* ------------------------------------------------------------------------ */
/* split.h / tags.h: breakpoint window, split detection, coordinates          */

static inline int is_tra(int svt) { return (5 <= svt) && (svt < 9); } /* tags.h:22-25 */
static inline int span_orient(int svt) { return is_tra(svt) ? svt - 5 : svt; } /* tags.h:33-40 */

typedef struct { /* tags.h:132-148 */
  int32_t svStartBeg, svStartEnd, svEndBeg, svEndEnd, svStart, svEnd, svt, chr, chr2;
} bpoint;

/* tags.h:151-172 */
static void init_breakpoint(const int64_t* chr_len, bpoint* bp, int32_t boundary, int svt) {
  if (is_tra(svt) || svt == 4) {
    bp->svStartBeg = imax(0, bp->svStart - boundary);
    bp->svStartEnd = imin((int32_t)(uint32_t)chr_len[bp->chr], bp->svStart + boundary);
    bp->svEndBeg = imax(0, bp->svEnd - boundary);
    bp->svEndEnd = imin((int32_t)(uint32_t)chr_len[bp->chr2], bp->svEnd + boundary);
  } else {
    bp->svStartBeg = imax(0, bp->svStart - boundary);
    bp->svStartEnd = imin(bp->svStart + boundary, (bp->svStart + bp->svEnd) / 2);
    bp->svEndBeg = imax((bp->svStart + bp->svEnd) / 2 + 1, bp->svEnd - boundary);
    bp->svEndEnd = imin((int32_t)(uint32_t)chr_len[bp->chr2], bp->svEnd + boundary);
  }
}

typedef struct {
  char* d;
  size_t n, cap;
} sbuf;
static void sb_init(sbuf* s) { s->d = (char*)malloc(64); s->n = 0; s->cap = 64; }
static void sb_reserve(sbuf* s, size_t extra) {
  if (s->n + extra + 1 > s->cap) {
    while (s->n + extra + 1 > s->cap) s->cap *= 2;
    s->d = (char*)realloc(s->d, s->cap);
  }
}
/* append upper(ref[beg,end)) -- boost::to_upper_copy(std::string(ref+beg, ref+end)) */
static void sb_upper(sbuf* s, const char* ref, int32_t beg, int32_t end) {
  if (end <= beg) return;
  sb_reserve(s, (size_t)(end - beg));
  for (int32_t i = beg; i < end; ++i) s->d[s->n++] = (char)toupper((unsigned char)ref[i]);
}
static void sb_append(sbuf* s, const char* p, size_t n) {
  sb_reserve(s, n);
  memcpy(s->d + s->n, p, n);
  s->n += n;
}
/* append the split.h:78-91 style reverse complement of upper(ref[beg,end)):
 * out[i] = comp(str[len-1-i]) for ACGTN, else str[i] (un-reversed) */
static void sb_upper_rc(sbuf* s, const char* ref, int32_t beg, int32_t end) {
  if (end <= beg) return;
  size_t L = (size_t)(end - beg);
  sb_reserve(s, L);
  char* o = s->d + s->n;
  for (size_t i = 0; i < L; ++i) {
    char fwd = (char)toupper((unsigned char)ref[beg + (int32_t)i]);
    char r = (char)toupper((unsigned char)ref[beg + (int32_t)(L - 1 - i)]);
    switch (r) {
      case 'A': o[i] = 'T'; break;
      case 'C': o[i] = 'G'; break;
      case 'G': o[i] = 'C'; break;
      case 'T': o[i] = 'A'; break;
      case 'N': o[i] = 'N'; break;
      default: o[i] = fwd; break;
    }
  }
  s->n += L;
}

/* _getSVRef  src/split.h:70-163.  part1 = second-chromosome part (BND). */
static void get_sv_ref(const dellyhip_params* c, const char* ref, const bpoint* r, int refIndex,
                       int svt, const sbuf* part1, sbuf* out) {
  out->n = 0;
  if (is_tra(svt)) {
    int ct = span_orient(svt);
    if (r->chr == refIndex) {
      if (ct == 0 || ct == 2) {
        sb_upper(out, ref, r->svStartBeg, r->svStartEnd);
        sb_append(out, part1->d, part1->n);
      } else if (ct == 1) {
        sb_upper_rc(out, ref, r->svStartBeg, r->svStartEnd);
        sb_append(out, part1->d, part1->n);
      } else {
        sb_append(out, part1->d, part1->n);
        sb_upper(out, ref, r->svStartBeg, r->svStartEnd);
      }
    } else {
      if (ct == 0) sb_upper_rc(out, ref, r->svEndBeg, r->svEndEnd);
      else sb_upper(out, ref, r->svEndBeg, r->svEndEnd);
    }
  } else if (svt == 2) {
    if (r->svEnd - r->svStart <= c->indelsize) sb_upper(out, ref, r->svStartBeg, r->svEndEnd);
    else {
      sb_upper(out, ref, r->svStartBeg, r->svStartEnd);
      sb_upper(out, ref, r->svEndBeg, r->svEndEnd);
    }
  } else if (svt == 4) {
    sb_upper(out, ref, r->svStartBeg, r->svEndEnd);
  } else if (svt == 3) {
    sb_upper(out, ref, r->svEndBeg, r->svEndEnd);
    sb_upper(out, ref, r->svStartBeg, r->svStartEnd);
  } else if (svt == 0) {
    int big = (r->svEnd - r->svStart) > c->min_cons_window;
    sb_upper(out, ref, r->svStartBeg, r->svStartEnd);
    if (big) sb_upper_rc(out, ref, r->svEndBeg, r->svEndEnd);
    else {
      sb_upper_rc(out, ref, r->svStart, r->svEndEnd);
      sb_upper(out, ref, r->svEnd, r->svEndEnd);
    }
  } else if (svt == 1) {
    int big = (r->svEnd - r->svStart) > c->min_cons_window;
    if (big) {
      sb_upper_rc(out, ref, r->svStartBeg, r->svStartEnd);
      sb_upper(out, ref, r->svEndBeg, r->svEndEnd);
    } else {
      sb_upper(out, ref, r->svStartBeg, r->svStart);
      sb_upper_rc(out, ref, r->svStartBeg, r->svEnd);
      sb_upper(out, ref, r->svEndBeg, r->svEndEnd);
    }
  }
}

typedef struct { /* split.h:15-25 */
  int32_t cStart, cEnd, rStart, rEnd, homLeft, homRight;
  float percId;
  uint32_t ma, mm;
} adesc;

/* _percentIdentity  split.h:282-316 */
static void percent_identity(const amat* al, int gS, int gE, adesc* ad) {
  int varSeen = 0, refSeen = 0, inGap = 0;
  uint32_t gapMM = 0, mm = 0, ma = 0;
  for (int j = 0; j < al->cols; ++j) {
    if (j < gS || j > gE) {
      char a0 = AT(*al, 0, j), a1 = AT(*al, 1, j);
      if (a0 != '-') varSeen = 1;
      if (a1 != '-') refSeen = 1;
      if (a0 == '-' || a1 == '-') {
        if (refSeen && varSeen) {
          if (!inGap) {
            inGap = 1;
            gapMM = 0;
          }
          gapMM += 1;
        }
      } else {
        if (inGap) {
          mm += gapMM;
          inGap = 0;
        }
        if (a0 == a1) ma += 1;
        else mm += 1;
      }
    }
  }
  ad->ma = ma;
  ad->mm = mm;
  ad->percId = (float)ma / (float)(ma + mm);
}

static char* dup_rev(const char* s, int n) {
  char* r = (char*)malloc((size_t)n + 1);
  for (int i = 0; i < n; ++i) r[i] = s[n - 1 - i];
  return r;
}

/* _findHomology  split.h:262-280 */
static void find_homology(const char* cons, int m, const char* ref, int n, adesc* ad, int svt) {
  if (svt == 4) {
    ad->homRight = dor_longest_homology(cons + ad->cStart, m - ad->cStart, ref + (ad->rEnd - 1), n - (ad->rEnd - 1), -1);
    int lc = imin(ad->cEnd - 1, m), lr = imin(ad->rStart, n);
    char* preC = dup_rev(cons, lc);
    char* preR = dup_rev(ref, lr);
    ad->homLeft = dor_longest_homology(preC, lc, preR, lr, -1);
    free(preC);
    free(preR);
  } else {
    ad->homRight = dor_longest_homology(cons + (ad->cEnd - 1), m - (ad->cEnd - 1), ref + ad->rStart, n - ad->rStart, -1);
    int lc = imin(ad->cStart, m), lr = imin(ad->rEnd - 1, n);
    char* preC = dup_rev(cons, lc);
    char* preR = dup_rev(ref, lr);
    ad->homLeft = dor_longest_homology(preC, lc, preR, lr, -1);
    free(preC);
    free(preR);
  }
}

/* _findSplit  split.h:319-375; gS/gE returned for the allele scan */
static int find_split(const dellyhip_params* c, const char* cons, int m, const char* ref, int n,
                      const amat* al, adesc* ad, int svt) {
  int gS = 0, gE = 0, refIndex = 0, varIndex = 0, gapStartRefIndex = 0, gapStartVarIndex = 0, a1 = 0;
  int inGap = 0;
  for (int j = 0; j < al->cols; ++j) {
    char c0 = AT(*al, 0, j), c1 = AT(*al, 1, j);
    if (c0 != '-') ++varIndex;
    if (c1 != '-') ++refIndex;
    if ((c0 == '-' || c1 == '-') && refIndex > 0 && varIndex > 0) {
      if (!inGap) {
        gapStartVarIndex = (c0 != '-') ? (varIndex - 1) : varIndex;
        gapStartRefIndex = (c1 != '-') ? (refIndex - 1) : refIndex;
        a1 = j;
        inGap = 1;
      }
    } else {
      int better = (svt == 4) ? ((varIndex - gapStartVarIndex) > (ad->cEnd - ad->cStart))
                              : ((refIndex - gapStartRefIndex) > (ad->rEnd - ad->rStart));
      if (inGap && better) {
        ad->rStart = gapStartRefIndex;
        ad->rEnd = refIndex;
        ad->cStart = gapStartVarIndex;
        ad->cEnd = varIndex;
        gS = a1;
        gE = j - 1;
      }
      inGap = 0;
    }
  }
  if (ad->rEnd <= ad->rStart) return 0;
  /* _validSRAlignment split.h:247-253 */
  if (svt == 4) {
    if (!(((ad->rEnd - ad->rStart) < 5) && ((ad->cEnd - ad->cStart) > 15))) return 0;
  } else {
    if (!(((ad->cEnd - ad->cStart) < 5) && ((ad->rEnd - ad->rStart) > 15))) return 0;
  }
  percent_identity(al, gS, gE, ad);
  if (ad->percId < c->flank_quality) return 0;
  find_homology(cons, m, ref, n, ad, svt);
  if ((ad->homLeft + c->minimum_flank_size > ad->cStart) || (varIndex < ad->cEnd + ad->homRight + c->minimum_flank_size)) return 0;
  if ((ad->homLeft + c->minimum_flank_size > ad->rStart) || (refIndex < ad->rEnd + ad->homRight + c->minimum_flank_size)) return 0;
  return 1;
}

/* _coordTransform  split.h:166-244 (unsigned int outputs) */
static int coord_transform(const dellyhip_params* c, uint64_t refsize, const bpoint* sv, const adesc* ad,
                           uint32_t* gs, uint32_t* ge, int svt) {
  int32_t annealed;
  if (is_tra(svt)) {
    int ct = span_orient(svt);
    if (ct == 0) {
      annealed = sv->svStartEnd - sv->svStartBeg;
      if (ad->rStart >= annealed || ad->rEnd < annealed) return 0;
      *gs = (uint32_t)(sv->svStartBeg + ad->rStart);
      *ge = (uint32_t)((uint64_t)(int64_t)sv->svEndBeg + (refsize - (uint64_t)(int64_t)ad->rEnd) + 1);
    } else if (ct == 1) {
      annealed = sv->svStartEnd - sv->svStartBeg;
      if (ad->rStart >= annealed || ad->rEnd < annealed) return 0;
      *gs = (uint32_t)(sv->svStartBeg + (annealed - ad->rStart) + 1);
      *ge = (uint32_t)(sv->svEndBeg + (ad->rEnd - annealed));
    } else if (ct == 2) {
      annealed = sv->svStartEnd - sv->svStartBeg;
      if (ad->rStart >= annealed || ad->rEnd < annealed) return 0;
      *gs = (uint32_t)(sv->svStartBeg + ad->rStart);
      *ge = (uint32_t)(sv->svEndBeg + (ad->rEnd - annealed));
    } else if (ct == 3) {
      annealed = sv->svEndEnd - sv->svEndBeg;
      if (ad->rStart >= annealed || ad->rEnd < annealed) return 0;
      *gs = (uint32_t)(sv->svStartBeg + (ad->rEnd - annealed));
      *ge = (uint32_t)(sv->svEndBeg + ad->rStart);
    } else return 0;
    return 1;
  }
  if (svt == 2) {
    if (sv->svEnd - sv->svStart > c->indelsize) {
      annealed = sv->svStartEnd - sv->svStartBeg;
      if (ad->rStart >= annealed || ad->rEnd < annealed) return 0;
      *gs = (uint32_t)(sv->svStartBeg + ad->rStart);
      *ge = (uint32_t)(sv->svEndBeg + (ad->rEnd - annealed));
    } else {
      *gs = (uint32_t)(sv->svStartBeg + ad->rStart);
      *ge = (uint32_t)(sv->svStartBeg + ad->rEnd);
    }
    return 1;
  } else if (svt == 3) {
    annealed = sv->svEndEnd - sv->svEndBeg;
    if (ad->rStart >= annealed || ad->rEnd < annealed) return 0;
    *gs = (uint32_t)(sv->svStartBeg + (ad->rEnd - annealed));
    *ge = (uint32_t)(sv->svEndBeg + ad->rStart);
    return 1;
  } else if (svt == 0) {
    annealed = sv->svStartEnd - sv->svStartBeg;
    if (ad->rStart >= annealed || ad->rEnd < annealed) return 0;
    if ((sv->svEnd - sv->svStart) > c->min_cons_window) {
      *gs = (uint32_t)(sv->svStartBeg + ad->rStart);
      *ge = (uint32_t)((uint64_t)(int64_t)sv->svEndBeg + (refsize - (uint64_t)(int64_t)ad->rEnd) + 1);
    } else {
      *gs = (uint32_t)(sv->svStartBeg + ad->rStart);
      *ge = (uint32_t)(sv->svEndEnd - (ad->rEnd - annealed));
    }
    return 1;
  } else if (svt == 1) {
    if ((sv->svEnd - sv->svStart) > c->min_cons_window) annealed = sv->svStartEnd - sv->svStartBeg;
    else annealed = (sv->svStart - sv->svStartBeg) + (sv->svEnd - sv->svStartBeg);
    if (ad->rStart >= annealed || ad->rEnd < annealed) return 0;
    *gs = (uint32_t)(sv->svStartBeg + (annealed - ad->rStart) + 1);
    *ge = (uint32_t)(sv->svEndBeg + (ad->rEnd - annealed));
    return 1;
  } else if (svt == 4) {
    *gs = (uint32_t)(sv->svStartBeg + ad->rStart);
    *ge = (uint32_t)(sv->svStartBeg + ad->rEnd);
    return 1;
  }
  return 1;
}

/* ------------------------------------------------------------------------ */
/* edlib (vendored third-party code of the reference: src/edlib.cpp, edlib    */
/* v1.2.x) restated as a plain unit-cost DP.  edlibAlign() runs Myers'        */
/* bit-vector algorithm inside an Ukkonen band with a doubling k; every       */
/* observable output (editDistance, endLocations[0], startLocations[0], the   */
/* op string of obtainAlignmentTraceback) is a function of the exact DP       */
/* matrix, which is what is computed here:                                    */
/*   - D[i][j], i over the query, j over the target; D[i][0] = i;             */
/*     D[0][j] = 0 (HW) or j (SHW, NW)                      edlib.cpp:573-575 */
/*   - HW/SHW end locations: columns of the last query row that attain the    */
/*     minimum, in increasing order; the column "before the target"           */
/*     (position -1) takes part only when the query length is not a multiple  */
/*     of 64, because it is observed through the W padding rows of the last   */
/*     block (c - W >= -1 needs W >= 1)                     edlib.cpp:653-691 */
/*   - HW start: SHW of the reversed query on the reversed target prefix,     */
/*     LAST optimal end                                     edlib.cpp:236-249 */
/*   - path: NW traceback on target[start..end], preferring "up" (INSERT,     */
/*     consumes a query letter), then "left" (DELETE), then the diagonal      */
/*                                                          edlib.cpp:1018-1125 */
/* Hirschberg mode (alignment data >= 1 MiB, edlib.cpp:1188-1191) breaks ties */
/* differently and is NOT restated: such calls return DOR_ED_LIMIT.           */

#define DOR_ED_LIMIT (-4)
enum { ED_NW = 0, ED_SHW = 1, ED_HW = 2 };
enum { OP_MATCH = 0, OP_INSERT = 1, OP_DELETE = 2, OP_MISMATCH = 3 };

typedef struct {
  int ed, num_loc, end_loc, start_loc; /* loc = -2: not set */
  unsigned char* aln;
  int aln_len;
} ed_res;

/* additional equalities (edlib.cpp:58-89): eq[a*256+b] != 0 when a and b count as equal; NULL = identity */
static inline int ed_neq(const unsigned char* eq, char a, char b) {
  if (a == b) return 0;
  return eq ? !eq[(size_t)(unsigned char)a * 256 + (unsigned char)b] : 1;
}

static int32_t* ed_fill_eq(const char* q, int qn, const char* t, int tn, int hw, const unsigned char* eq) {
  size_t W = (size_t)tn + 1;
  int32_t* D = (int32_t*)malloc(sizeof(int32_t) * (size_t)(qn + 1) * W);
  for (int j = 0; j <= tn; ++j) D[j] = hw ? 0 : j;
  for (int i = 1; i <= qn; ++i) {
    int32_t* r = D + (size_t)i * W;
    const int32_t* u = r - W;
    r[0] = i;
    for (int j = 1; j <= tn; ++j) {
      int v = u[j - 1] + ed_neq(eq, q[i - 1], t[j - 1]);
      if (u[j] + 1 < v) v = u[j] + 1;
      if (r[j - 1] + 1 < v) v = r[j - 1] + 1;
      r[j] = v;
    }
  }
  return D;
}

/* last column of the NW matrix: out[i] = distance(q[0..i), t[0..tn)), i = 0..qn (two rolling rows) */
static void ed_last_column(const char* q, int qn, const char* t, int tn, const unsigned char* eq, int32_t* out) {
  int32_t* prev = (int32_t*)malloc(sizeof(int32_t) * ((size_t)tn + 1));
  int32_t* cur = (int32_t*)malloc(sizeof(int32_t) * ((size_t)tn + 1));
  for (int j = 0; j <= tn; ++j) prev[j] = j;
  out[0] = tn;
  for (int i = 1; i <= qn; ++i) {
    cur[0] = i;
    for (int j = 1; j <= tn; ++j) {
      int v = prev[j - 1] + ed_neq(eq, q[i - 1], t[j - 1]);
      if (prev[j] + 1 < v) v = prev[j] + 1;
      if (cur[j - 1] + 1 < v) v = cur[j - 1] + 1;
      cur[j] = v;
    }
    out[i] = cur[tn];
    int32_t* x = prev; prev = cur; cur = x;
  }
  free(prev);
  free(cur);
}

/* obtainAlignmentTraceback (edlib.cpp:943-1143) on an exact NW matrix; appends the ops (forward order) */
static void ed_trace_append(const char* q, int qn, const char* t, int tn, const unsigned char* eq, unsigned char* out, int* outn) {
  int32_t* D = ed_fill_eq(q, qn, t, tn, 0, eq);
  size_t W = (size_t)tn + 1;
  unsigned char* ops = out + *outn;
  int L = 0, i = qn, j = tn;
  while (i > 0 && j > 0) {
    int cur = D[(size_t)i * W + j];
    if (D[(size_t)(i - 1) * W + j] + 1 == cur) { ops[L++] = OP_INSERT; --i; }
    else if (D[(size_t)i * W + j - 1] + 1 == cur) { ops[L++] = OP_DELETE; --j; }
    else { ops[L++] = (D[(size_t)(i - 1) * W + j - 1] == cur) ? OP_MATCH : OP_MISMATCH; --i; --j; }
  }
  while (j > 0) { ops[L++] = OP_DELETE; --j; }  /* query exhausted: edlib.cpp:1027-1031,1087-1091 */
  while (i > 0) { ops[L++] = OP_INSERT; --i; }  /* target exhausted: edlib.cpp:1057-1062,1079-1084 */
  for (int a = 0, b = L - 1; a < b; ++a, --b) { unsigned char x = ops[a]; ops[a] = ops[b]; ops[b] = x; }
  free(D);
  *outn += L;
}

/* obtainAlignment (edlib.cpp:1163-1201): traceback below 1 MiB of alignment data, otherwise
 * obtainAlignmentHirschberg (:1220-1389): split the target in the middle, take the FIRST query
 * index (ascending; then the two boundary cases) whose left + right scores add up to the optimum,
 * recurse on the upper-left and lower-right rectangles, concatenate.  edlib searches only inside
 * its Ukkonen bands; cells on optimal paths are always inside them and exact, so the exact score
 * columns give the same index. */
static void ed_path_append(const char* q, int qn, const char* t, int tn, const unsigned char* eq, unsigned char* out, int* outn) {
  if (qn == 0 || tn == 0) {
    memset(out + *outn, qn == 0 ? OP_DELETE : OP_INSERT, (size_t)(qn + tn));
    *outn += qn + tn;
    return;
  }
  long long blocks = (qn + 63) / 64;
  long long sz = (2ll * 8 + 4) * blocks * tn + 2ll * 4 * tn;
  if (sz < 1024 * 1024) {
    ed_trace_append(q, qn, t, tn, eq, out, outn);
    return;
  }
  const int lw = tn / 2, rw = tn - lw;
  int32_t* left = (int32_t*)malloc(sizeof(int32_t) * ((size_t)qn + 1));
  int32_t* rightr = (int32_t*)malloc(sizeof(int32_t) * ((size_t)qn + 1));
  char* rq = (char*)malloc((size_t)qn + 1);
  char* rt = (char*)malloc((size_t)rw + 1);
  for (int i = 0; i < qn; ++i) rq[i] = q[qn - 1 - i];
  for (int j = 0; j < rw; ++j) rt[j] = t[tn - 1 - j];
  ed_last_column(q, qn, t, lw, eq, left);      /* left[i]  : q[0..i) vs t[0..lw)      */
  ed_last_column(rq, qn, rt, rw, eq, rightr);  /* rightr[k]: q[qn-k..qn) vs t[lw..tn) */
  int bestScore = 1 << 30;
  for (int i = 0; i <= qn; ++i) if (left[i] + rightr[qn - i] < bestScore) bestScore = left[i] + rightr[qn - i];
  int ul = -2; /* number of query letters in the upper-left part */
  for (int qi = 0; qi <= qn - 2; ++qi)          /* queryIdx = qi: left part holds qi + 1 letters */
    if (left[qi + 1] + rightr[qn - (qi + 1)] == bestScore) { ul = qi + 1; break; }
  if (ul == -2 && left[0] + rightr[qn] == bestScore) ul = 0;        /* queryIdx = -1 */
  if (ul == -2 && left[qn] + rightr[0] == bestScore) ul = qn;       /* queryIdx = qn - 1 */
  free(left); free(rightr); free(rq); free(rt);
  ed_path_append(q, ul, t, lw, eq, out, outn);
  ed_path_append(q + ul, qn - ul, t + lw, rw, eq, out, outn);
}

/* edlibAlign(query, target, edlibNewAlignConfig(-1, mode, task, NULL, 0)); task 0 DISTANCE, 1 LOC, 2 PATH */
static int ed_align_eq(const char* q, int qn, const char* t, int tn, int mode, int task, const unsigned char* eq, ed_res* r) {
  r->ed = -1; r->num_loc = 0; r->end_loc = r->start_loc = -2; r->aln = NULL; r->aln_len = 0;
  if (qn == 0 || tn == 0) { /* edlib.cpp:160-178 */
    if (mode == ED_NW) { r->ed = imax(qn, tn); r->end_loc = tn - 1; }
    else { r->ed = qn; r->end_loc = -1; }
    r->num_loc = 1;
    return 0;
  }
  const int pad = ((qn + 63) / 64) * 64 - qn; /* W */
  const int j0 = pad >= 1 ? 0 : 1;
  int32_t* D = ed_fill_eq(q, qn, t, tn, mode == ED_HW, eq);
  const int32_t* last = D + (size_t)qn * ((size_t)tn + 1);
  if (mode == ED_NW) {
    r->ed = last[tn];
    r->end_loc = tn - 1;
    r->num_loc = 1;
  } else {
    int best = last[j0];
    for (int j = j0; j <= tn; ++j) if (last[j] < best) best = last[j];
    r->ed = best;
    for (int j = tn; j >= j0; --j) if (last[j] == best) { r->end_loc = j - 1; r->num_loc++; }
  }
  free(D);
  if (task == 0) return 0;
  if (mode == ED_HW) {
    if (r->end_loc == -1) r->start_loc = 0; /* edlib.cpp:222-235 */
    else {
      int tl = r->end_loc + 1;
      char* rq = (char*)malloc((size_t)qn + 1);
      char* rt = (char*)malloc((size_t)tl + 1);
      for (int i = 0; i < qn; ++i) rq[i] = q[qn - 1 - i];
      for (int j = 0; j < tl; ++j) rt[j] = t[r->end_loc - j];
      int32_t* R = ed_fill_eq(rq, qn, rt, tl, 0, eq);
      const int32_t* rl = R + (size_t)qn * ((size_t)tl + 1);
      int best = rl[j0], lastj = j0;
      for (int j = j0; j <= tl; ++j) {
        if (rl[j] < best) best = rl[j];
      }
      for (int j = j0; j <= tl; ++j) if (rl[j] == best) lastj = j;
      r->start_loc = r->end_loc - (lastj - 1);
      free(R); free(rq); free(rt);
    }
  } else r->start_loc = 0;
  if (task == 1) return 0;
  {
    const int s0 = r->start_loc, tl = r->end_loc - s0 + 1;
    if (tl == 0) { /* obtainAlignment edlib.cpp:1169-1176 */
      r->aln_len = qn;
      r->aln = (unsigned char*)malloc((size_t)qn + 1);
      memset(r->aln, OP_INSERT, (size_t)qn);
      return 0;
    }
    r->aln = (unsigned char*)malloc((size_t)qn + (size_t)tl + 1);
    r->aln_len = 0;
    ed_path_append(q, qn, t + s0, tl, eq, r->aln, &r->aln_len);
  }
  return 0;
}

static int ed_align(const char* q, int qn, const char* t, int tn, int mode, int task, ed_res* r) {
  return ed_align_eq(q, qn, t, tn, mode, task, NULL, r);
}

/* the 20 extended-IUPAC pairs of msaEdlib / msaWfa (src/assemble.h:425, :660) as an equality table */
static const unsigned char* iupac_equalities(void) {
  static unsigned char tab[256 * 256];
  static int ready = 0;
  if (!ready) {
    static const char pairs[20][2] = {{'M', 'A'}, {'M', 'C'}, {'R', 'A'}, {'R', 'G'}, {'W', 'A'}, {'W', 'T'}, {'B', 'A'},
                                      {'B', '-'}, {'S', 'C'}, {'S', 'G'}, {'Y', 'C'}, {'Y', 'T'}, {'D', 'C'}, {'D', '-'},
                                      {'K', 'G'}, {'K', 'T'}, {'E', 'G'}, {'E', '-'}, {'F', 'T'}, {'F', '-'}};
    for (int i = 0; i < 20; ++i) {
      tab[(size_t)(unsigned char)pairs[i][0] * 256 + (unsigned char)pairs[i][1]] = 1;
      tab[(size_t)(unsigned char)pairs[i][1] * 256 + (unsigned char)pairs[i][0] ] = 1;
    }
    ready = 1;
  }
  return tab;
}

/* mode | 16: with the extended-IUPAC additional equalities */
int dor_edlib_align(const char* q, int qn, const char* t, int tn, int mode, int task, int* out,
                    unsigned char* aln, int cap) {
  ed_res r;
  const unsigned char* eq = (mode & 16) ? iupac_equalities() : NULL;
  mode &= 15;
  int rc = ed_align_eq(q, qn, t, tn, mode, task, eq, &r);
  if (rc) return rc;
  out[0] = r.ed; out[1] = r.num_loc; out[2] = r.end_loc; out[3] = r.start_loc;
  int L = r.aln_len;
  if (r.aln && L <= cap) memcpy(aln, r.aln, (size_t)L);
  free(r.aln);
  return L;
}

/* infixStart / infixEnd  src/util.h:86-99 */
static uint32_t infix_start(const ed_res* c) {
  int32_t tIdx = c->end_loc;
  for (int i = 0; i < c->aln_len; ++i) if (c->aln[i] != OP_INSERT) --tIdx;
  return tIdx >= 0 ? (uint32_t)(tIdx + 1) : 0u;
}

/* editDistanceVec  src/split.h:377-405 */
static void edit_distance_vec(const char* sI, int nI, const char* sJ, const ed_res* c, uint32_t* dist) {
  for (int i = 0; i < nI; ++i) dist[i] = 0;
  int32_t tIdx = -1, qIdx = -1;
  uint32_t ed = 0;
  for (int j = 0; j < c->aln_len; ++j) {
    if (c->aln[j] == OP_DELETE) { ++tIdx; ++ed; }
    else if (c->aln[j] == OP_INSERT) { ++qIdx; ++ed; dist[qIdx] = ed; }
    else { ++tIdx; ++qIdx; if (sI[qIdx] != sJ[tIdx]) ++ed; dist[qIdx] = ed; }
  }
}

/* splitAlign  src/split.h:480-538 + the row swap of _consRefAlignment :546-552.
 * Returns 1 ok, 0 false, <0 error; *out rows: [0] consensus, [1] reference.
 * internals[5] = {csStart, csEnd, bestJoin, leftEnd, rightStart} (-1 when not reached). */
static int split_align(const char* cons, int m, const char* ref, int n, amat* out, int* internals) {
  for (int i = 0; i < 5; ++i) internals[i] = -1;
  out->d = NULL; out->rows = 2; out->cols = 0;
  if (n < 3) return DOR_ED_LIMIT; /* the reference indexes distRev[n-2] and loops to size()-1 */
  ed_res c;
  int rc;
  if ((rc = ed_align(ref, n / 3, cons, m, ED_HW, 2, &c))) return rc;
  uint32_t csStart = infix_start(&c);
  free(c.aln);
  int so = (int)(2 * (size_t)n / 3);
  if ((rc = ed_align(ref + so, n - so, cons, m, ED_HW, 2, &c))) return rc;
  uint32_t csEnd = (uint32_t)c.end_loc;
  free(c.aln);
  internals[0] = (int)csStart;
  internals[1] = (int)csEnd;
  if (csStart >= csEnd) return 0;
  /* std::string::substr(pos, len): len clamped to size - pos */
  uint32_t csl = csEnd - csStart;
  if (csl > (uint32_t)m - csStart) csl = (uint32_t)m - csStart;
  const char* cs = cons + csStart;
  uint32_t* distFwd = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)n);
  uint32_t* distRev = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)n);
  if ((rc = ed_align(ref, n, cs, (int)csl, ED_SHW, 2, &c))) { free(distFwd); free(distRev); return rc; }
  edit_distance_vec(ref, n, cs, &c, distFwd);
  free(c.aln);
  char* refRev = (char*)malloc((size_t)n + 1);
  char* csRev = (char*)malloc((size_t)csl + 1);
  memcpy(refRev, ref, (size_t)n);
  memcpy(csRev, cs, (size_t)csl);
  dor_reverse_complement(refRev, n);
  dor_reverse_complement(csRev, (int)csl);
  if ((rc = ed_align(refRev, n, csRev, (int)csl, ED_SHW, 2, &c))) { free(distFwd); free(distRev); free(refRev); free(csRev); return rc; }
  edit_distance_vec(refRev, n, csRev, &c, distRev);
  free(c.aln);
  free(refRev);
  free(csRev);
  uint32_t bestJoin = 0;
  for (uint32_t i = 1; i < (uint32_t)n - 1; ++i)
    if (distFwd[i] + distRev[n - i - 2] < distFwd[bestJoin] + distRev[n - bestJoin - 2]) bestJoin = i;
  free(distFwd);
  free(distRev);
  internals[2] = (int)bestJoin;
  ed_res cl, cr;
  if ((rc = ed_align(ref, (int)bestJoin + 1, cons, m, ED_HW, 2, &cl))) return rc;
  uint32_t leftEnd = (uint32_t)cl.end_loc;
  if ((rc = ed_align(ref + bestJoin + 1, n - (int)bestJoin - 1, cons, m, ED_HW, 2, &cr))) { free(cl.aln); return rc; }
  uint32_t rightStart = infix_start(&cr);
  internals[3] = (int)leftEnd;
  internals[4] = (int)rightStart;
  if (leftEnd + 15u >= rightStart) { free(cl.aln); free(cr.aln); return 0; }
  /* glueAlignment(svRefStr, cons, gaplen, HW, ...)  split.h:407-477: row 0 query (ref), row 1 target (cons) */
  uint32_t gaplen = rightStart - leftEnd - 1u;
  int32_t tIdx = cl.end_loc, qIdx = -1;
  uint32_t missingStart = 0;
  for (int i = 0; i < cl.aln_len; ++i) if (cl.aln[i] != OP_INSERT) --tIdx;
  if (tIdx >= 0) missingStart = (uint32_t)tIdx + 1u;
  uint32_t missingEnd = (uint32_t)cr.end_loc;
  if (missingEnd < (uint32_t)m) missingEnd = (uint32_t)m - missingEnd - 1u;
  uint64_t total = (uint64_t)missingStart + (uint64_t)cl.aln_len + gaplen + (uint64_t)cr.aln_len + missingEnd;
  if (total > (uint64_t)m + (uint64_t)n + 8) { free(cl.aln); free(cr.aln); return DOR_ED_LIMIT; }
  amat g = amat_new(2, (int)total);
  uint32_t o = 0;
  for (uint32_t j = 0; j < missingStart; ++j) { AT(g, 1, j) = cons[j]; AT(g, 0, j) = '-'; }
  o = missingStart;
  for (int j = 0; j < cl.aln_len; ++j) {
    if (cl.aln[j] == OP_INSERT) AT(g, 1, o + j) = '-';
    else AT(g, 1, o + j) = cons[++tIdx];
  }
  for (int j = 0; j < cl.aln_len; ++j) {
    if (cl.aln[j] == OP_DELETE) AT(g, 0, o + j) = '-';
    else AT(g, 0, o + j) = ref[++qIdx];
  }
  o += (uint32_t)cl.aln_len;
  for (uint32_t j = 0; j < gaplen; ++j) { AT(g, 0, o + j) = '-'; AT(g, 1, o + j) = cons[++tIdx]; }
  o += gaplen;
  for (int j = 0; j < cr.aln_len; ++j) {
    if (cr.aln[j] == OP_INSERT) AT(g, 1, o + j) = '-';
    else AT(g, 1, o + j) = cons[++tIdx];
  }
  for (int j = 0; j < cr.aln_len; ++j) {
    if (cr.aln[j] == OP_DELETE) AT(g, 0, o + j) = '-';
    else AT(g, 0, o + j) = ref[++qIdx];
  }
  o += (uint32_t)cr.aln_len;
  for (uint32_t j = 0; j < missingEnd; ++j) { AT(g, 1, o + j) = cons[++tIdx]; AT(g, 0, o + j) = '-'; }
  free(cl.aln);
  free(cr.aln);
  /* swap rows: split.h:548-552 */
  for (int j = 0; j < g.cols; ++j) { char x = AT(g, 0, j); AT(g, 0, j) = AT(g, 1, j); AT(g, 1, j) = x; }
  *out = g;
  return 1;
}

int dor_split_align(const char* cons, int m, const char* ref, int n, char* rows, int cap, int* len, int* internals) {
  amat g;
  int in5[5];
  int rc = split_align(cons, m, ref, n, &g, in5);
  if (internals) memcpy(internals, in5, sizeof(in5));
  *len = (rc == 1) ? g.cols : 0;
  if (rc != 1) return rc;
  if (g.cols > cap) { amat_free(&g); return -1; }
  memcpy(rows, g.d, (size_t)g.cols);
  memcpy(rows + cap, g.d + g.cols, (size_t)g.cols);
  amat_free(&g);
  return 1;
}

/* ------------------------------------------------------------------------ */
/* msaEdlib  src/assemble.h:383-473 (long-read consensus, `delly lr` non-INS)  */

/* consensusEdlib  src/assemble.h:198-259: per column the majority of A,C,G,T,'-' or, when the
 * runner-up has at least half the majority's count, the two-letter extended-IUPAC code */
static void consensus_edlib(const amat* al, char* cons) {
  for (int j = 0; j < al->cols; ++j) {
    int count[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < al->rows; ++i) {
      char ch = AT(*al, i, j);
      if (ch == 'A' || ch == 'a') ++count[0];
      else if (ch == 'C' || ch == 'c') ++count[1];
      else if (ch == 'G' || ch == 'g') ++count[2];
      else if (ch == 'T' || ch == 't') ++count[3];
      else ++count[4];
    }
    uint32_t maxIdx = 0, sndIdx = 1;
    if (count[maxIdx] < count[sndIdx]) { maxIdx = 1; sndIdx = 0; }
    for (uint32_t i = 2; i < 5; ++i) {
      if (count[i] > count[maxIdx]) { sndIdx = maxIdx; maxIdx = i; }
      else if (count[i] > count[sndIdx]) sndIdx = i;
    }
    if (2 * count[sndIdx] < count[maxIdx]) {
      cons[j] = maxIdx < 4 ? "ACGT"[maxIdx] : '-';
    } else {
      uint32_t k1 = maxIdx, k2 = sndIdx;
      if (k1 > k2) { k1 = sndIdx; k2 = maxIdx; }
      static const char code[5][5] = {{'-', 'M', 'R', 'W', 'B'}, {'-', '-', 'S', 'Y', 'D'}, {'-', '-', '-', 'K', 'E'},
                                      {'-', '-', '-', '-', 'F'}, {'-', '-', '-', '-', '-'}};
      cons[j] = code[k1][k2];
    }
  }
}

/* convertAlignment(query, align, EDLIB_MODE_NW, cigar)  src/assemble.h:24-88 */
static amat convert_alignment_nw(const char* query, const amat* in, const unsigned char* ops, int nops) {
  amat out = amat_new(in->rows + 1, nops);
  int tIdx = -1, qIdx = -1;
  for (int j = 0; j < nops; ++j) {
    if (ops[j] == OP_INSERT) {
      for (int r = 0; r < in->rows; ++r) AT(out, r, j) = '-';
    } else {
      ++tIdx;
      for (int r = 0; r < in->rows; ++r) AT(out, r, j) = AT(*in, r, tIdx);
    }
  }
  for (int j = 0; j < nops; ++j) AT(out, in->rows, j) = (ops[j] == OP_DELETE) ? '-' : query[++qIdx];
  return out;
}

static int cmp_int(const void* a, const void* b) { int x = *(const int*)a, y = *(const int*)b; return (x > y) - (x < y); }
static int cmp_pair(const void* a, const void* b) {
  const int32_t* x = (const int32_t*)a; const int32_t* y = (const int32_t*)b;
  if (x[0] != y[0]) return (x[0] > y[0]) - (x[0] < y[0]);
  return (x[1] > y[1]) - (x[1] < y[1]);
}

static int msa_edlib_core(const dellyhip_params* p, int n, const char* blob, const uint64_t* off, char** cs_out, int* cs_len) {
  const unsigned char* eq = iupac_equalities();
  int32_t* edit = (int32_t*)calloc((size_t)n * n, sizeof(int32_t));
  int maxlen = 0;
  for (int i = 0; i < n; ++i) maxlen = imax(maxlen, (int)(off[i + 1] - off[i]));
  int32_t* col = (int32_t*)malloc(sizeof(int32_t) * ((size_t)maxlen + 2));
  for (int i = 0; i < n; ++i)
    for (int j = i + 1; j < n; ++j) {
      const int li = (int)(off[i + 1] - off[i]), lj = (int)(off[j + 1] - off[j]);
      int d;
      if (li == 0 || lj == 0) d = imax(li, lj);
      else { ed_last_column(blob + off[i], li, blob + off[j], lj, NULL, col); d = col[li]; }
      edit[i * n + j] = edit[j * n + i] = d;
    }
  free(col);
  /* medoid: smallest median distance, first wins (assemble.h:397-408) */
  uint32_t bestIdx = 0;
  int32_t bestVal = (int32_t)(off[1] - off[0]);
  int* dist = (int*)malloc(sizeof(int) * (size_t)n);
  for (int i = 0; i < n; ++i) {
    for (int j = 0; j < n; ++j) dist[j] = edit[i * n + j];
    qsort(dist, (size_t)n, sizeof(int), cmp_int);
    if (dist[n / 2] < bestVal) { bestVal = dist[n / 2]; bestIdx = (uint32_t)i; }
  }
  free(dist);
  /* order by distance to the medoid, drop the poorest 20 % (keep >= 3) (:410-424) */
  int32_t* qs = (int32_t*)malloc(sizeof(int32_t) * 2 * (size_t)n);
  int nq = 0;
  qs[0] = 0; qs[1] = (int32_t)bestIdx; nq = 1;
  for (int j = 0; j < n; ++j)
    if ((uint32_t)j != bestIdx) { qs[2 * nq] = edit[bestIdx * n + j]; qs[2 * nq + 1] = j; ++nq; }
  qsort(qs, (size_t)nq, 2 * sizeof(int32_t), cmp_pair);
  uint32_t lastIdx = (uint32_t)(0.8 * nq);
  if (lastIdx < 3) lastIdx = 3;
  int nsel = 0;
  int* sel = (int*)malloc(sizeof(int) * (size_t)n);
  for (uint32_t i = 0; i < (uint32_t)nq && i < lastIdx; ++i) sel[nsel++] = qs[2 * i + 1];
  free(qs);
  free(edit);
  /* progressive alignment against the running 2-allele consensus (:426-447) */
  const int l0 = (int)(off[sel[0] + 1] - off[sel[0]]);
  amat al = amat_new(1, l0);
  memcpy(al.d, blob + off[sel[0]], (size_t)l0);
  for (int i = 1; i < nsel; ++i) {
    char* astr = (char*)malloc((size_t)al.cols + 1);
    consensus_edlib(&al, astr);
    const char* q = blob + off[sel[i]];
    const int qn = (int)(off[sel[i] + 1] - off[sel[i]]);
    ed_res c;
    ed_align_eq(q, qn, astr, al.cols, ED_NW, 2, eq, &c);
    amat nx = convert_alignment_nw(q, &al, c.aln, c.aln_len);
    free(c.aln);
    free(astr);
    amat_free(&al);
    al = nx;
  }
  free(sel);
  /* consensus(c, align, gapped, cs) (src/msa.h:111-173), then trim 5 % per side, <= 50 (:465-469) */
  char* cs = (char*)malloc((size_t)al.cols + 1);
  int L = consensus_core(p, &al, cs);
  int32_t trim = (int32_t)(0.05 * L);
  if (trim > 50) trim = 50;
  int32_t len = L - 2 * trim;
  if (len > 100) { memmove(cs, cs + trim, (size_t)len); L = len; }
  const int rows = al.rows;
  amat_free(&al);
  *cs_out = cs;
  *cs_len = L;
  return rows;
}

int dor_msa_edlib(const dellyhip_params* p, int n_reads, const char* blob, const uint64_t* off, char* cs, int cap, int* cs_len) {
  char* c = NULL;
  int L = 0;
  int rows = msa_edlib_core(p, n_reads, blob, off, &c, &L);
  *cs_len = L;
  if (L <= cap) memcpy(cs, c, (size_t)L);
  free(c);
  return rows;
}

/* ------------------------------------------------------------------------ */
/* msaWfa  src/assemble.h:547-726 (long-read consensus of insertion junctions) */

#define DOR_KMER 7                     /* DELLY_KMER, src/tags.h:19 */
#define DOR_KTAB 65536                 /* std::pow(4, DELLY_KMER + 1) */
#define DOR_DUP 0xffffffffu            /* DELLY_DUPLICATE, src/tags.h:15 */

static uint32_t char_to_int(char c) { /* assemble.h:475-498 */
  switch (c) {
    case 'A': case 'B': return 0;
    case 'C': case 'D': return 1;
    case 'G': case 'E': return 2;
    case 'T': case 'F': return 3;
  }
  return 0;
}

/* fillKmerTable  assemble.h:501-520 (uint32 arithmetic as written) */
static void fill_kmer_table(const char* s, uint32_t len, uint32_t* kmerpos) {
  memset(kmerpos, 0, sizeof(uint32_t) * DOR_KTAB);
  uint32_t hash = 0;
  for (uint32_t ki = 0; ki < len && ki < DOR_KMER; ++ki) { hash *= 4; hash += char_to_int(s[ki]); }
  for (uint32_t ki = DOR_KMER; ki < len; ++ki) {
    if (kmerpos[hash]) kmerpos[hash] = DOR_DUP;
    else kmerpos[hash] = ki - DOR_KMER + 1;
    hash -= char_to_int(s[ki - DOR_KMER]) * 4 * 4 * 4 * 4 * 4 * 4;
    hash *= 4;
    hash += char_to_int(s[ki]);
  }
  if (kmerpos[hash]) kmerpos[hash] = DOR_DUP;
  else kmerpos[hash] = len - DOR_KMER + 1;
}

/* bestDiagonal  assemble.h:522-545 */
static int32_t best_diagonal(const uint32_t* hitI, const uint32_t* hitJ, uint32_t lenI, uint32_t lenJ) {
  uint32_t dn = lenI + lenJ;
  uint32_t* diag = (uint32_t*)calloc((size_t)dn + 1, sizeof(uint32_t));
  for (uint32_t k = 0; k < DOR_KTAB; ++k)
    if (hitI[k] && hitJ[k] && hitI[k] != DOR_DUP && hitJ[k] != DOR_DUP) ++diag[lenJ + hitI[k] - hitJ[k]];
  uint32_t window = 20, windowVal = 0;
  for (uint32_t d = 0; d < dn && d < window; ++d) windowVal += diag[d];
  uint32_t bestDiag = window / 2, bestWindowVal = windowVal;
  for (uint32_t d = window; d < dn; ++d) {
    windowVal -= diag[d - window];
    windowVal += diag[d];
    if (windowVal > bestWindowVal) { bestWindowVal = windowVal; bestDiag = d - window / 2; }
  }
  free(diag);
  return (int)bestDiag - (int)lenJ;
}

/* buildSuperstring  assemble.h:90-133 */
static char* build_superstring(const char* seqI, const char* seqJ, const ed_res* cg, uint32_t preI, uint32_t postI,
                               uint32_t preJ, uint32_t postJ, size_t cap, int* outn) {
  char* out = (char*)malloc(cap + 1);
  int n = 0;
  int32_t iIdx = 0, jIdx = 0;
  int firstSeq = 0;
  if (preI > preJ) {
    firstSeq = 1;
    for (uint32_t j = 0; j < preI; ++j) out[n++] = seqI[iIdx++];
    for (uint32_t j = 0; j < preJ; ++j) ++jIdx;
  } else {
    for (uint32_t j = 0; j < preI; ++j) ++iIdx;
    for (uint32_t j = 0; j < preJ; ++j) out[n++] = seqJ[jIdx++];
  }
  int32_t bp = cg->aln_len / 2;
  for (int32_t j = 0; j < cg->aln_len; ++j) {
    if (bp == j) firstSeq = !firstSeq;
    if (cg->aln[j] == OP_DELETE) { if (!firstSeq) out[n++] = seqJ[jIdx]; ++jIdx; }
    else if (cg->aln[j] == OP_INSERT) { if (firstSeq) out[n++] = seqI[iIdx]; ++iIdx; }
    else { out[n++] = firstSeq ? seqI[iIdx] : seqJ[jIdx]; ++iIdx; ++jIdx; }
  }
  if (postI > postJ) { for (uint32_t j = 0; j < postI; ++j) out[n++] = seqI[iIdx++]; }
  else { for (uint32_t j = 0; j < postJ; ++j) out[n++] = seqJ[jIdx++]; }
  *outn = n;
  return out;
}

/* consensusWfa  assemble.h:262-336: like consensusEdlib but only rows that span the column vote */
static void consensus_wfa(const amat* al, char* cons) {
  uint32_t* rs = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)al->rows);
  uint32_t* re = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)al->rows);
  for (int i = 0; i < al->rows; ++i) {
    rs[i] = (uint32_t)al->cols; re[i] = 0;
    for (int j = 0; j < al->cols; ++j)
      if (AT(*al, i, j) != '-') { if ((uint32_t)j < rs[i]) rs[i] = (uint32_t)j; if ((uint32_t)j > re[i]) re[i] = (uint32_t)j; }
  }
  for (int j = 0; j < al->cols; ++j) {
    int count[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < al->rows; ++i) {
      if ((uint32_t)j >= rs[i] && (uint32_t)j <= re[i]) {
        char ch = AT(*al, i, j);
        if (ch == 'A' || ch == 'a') ++count[0];
        else if (ch == 'C' || ch == 'c') ++count[1];
        else if (ch == 'G' || ch == 'g') ++count[2];
        else if (ch == 'T' || ch == 't') ++count[3];
        else ++count[4];
      }
    }
    uint32_t maxIdx = 0, sndIdx = 1;
    if (count[maxIdx] < count[sndIdx]) { maxIdx = 1; sndIdx = 0; }
    for (uint32_t i = 2; i < 5; ++i) {
      if (count[i] > count[maxIdx]) { sndIdx = maxIdx; maxIdx = i; }
      else if (count[i] > count[sndIdx]) sndIdx = i;
    }
    if (2 * count[sndIdx] < count[maxIdx]) cons[j] = maxIdx < 4 ? "ACGT"[maxIdx] : '-';
    else {
      uint32_t k1 = maxIdx, k2 = sndIdx;
      if (k1 > k2) { k1 = sndIdx; k2 = maxIdx; }
      static const char code[5][5] = {{'-', 'M', 'R', 'W', 'B'}, {'-', '-', 'S', 'Y', 'D'}, {'-', '-', '-', 'K', 'E'},
                                      {'-', '-', '-', '-', 'F'}, {'-', '-', '-', '-', '-'}};
      cons[j] = code[k1][k2];
    }
  }
  free(rs); free(re);
}

/* convertAlignment(query, align, EDLIB_MODE_HW, cigar)  assemble.h:24-88 */
static amat convert_alignment_hw(const char* query, const amat* in, const ed_res* cg) {
  int32_t tIdx = cg->end_loc, qIdx = -1;
  uint32_t missingEnd = 0, missingStart = 0;
  if (tIdx < in->cols) missingEnd = (uint32_t)(in->cols - tIdx - 1);
  for (int i = 0; i < cg->aln_len; ++i) if (cg->aln[i] != OP_INSERT) --tIdx;
  if (tIdx >= 0) missingStart = (uint32_t)tIdx + 1;
  const int rows = in->rows;
  amat out = amat_new(rows + 1, (int)(missingStart + (uint32_t)cg->aln_len + missingEnd));
  for (uint32_t j = 0; j < missingStart; ++j) {
    for (int r = 0; r < rows; ++r) AT(out, r, j) = AT(*in, r, j);
    AT(out, rows, j) = '-';
  }
  for (int j = 0; j < cg->aln_len; ++j) {
    if (cg->aln[j] == OP_INSERT) { for (int r = 0; r < rows; ++r) AT(out, r, j + missingStart) = '-'; }
    else { ++tIdx; for (int r = 0; r < rows; ++r) AT(out, r, j + missingStart) = AT(*in, r, tIdx); }
  }
  for (int j = 0; j < cg->aln_len; ++j) AT(out, rows, j + missingStart) = (cg->aln[j] == OP_DELETE) ? '-' : query[++qIdx];
  for (uint32_t j = (uint32_t)cg->aln_len + missingStart; j < (uint32_t)cg->aln_len + missingStart + missingEnd; ++j) {
    ++tIdx;
    for (int r = 0; r < rows; ++r) AT(out, r, j) = AT(*in, r, tIdx);
    AT(out, rows, j) = '-';
  }
  return out;
}

/* _trimConsensus  assemble.h:338-365; cs is modified in place, returns the new length */
static int trim_consensus(const char* prefix, int pn, const char* suffix, int sn, char* cs, int L) {
  char* prev = (char*)malloc((size_t)pn + 1);
  memcpy(prev, prefix, (size_t)pn);
  dor_reverse_complement(prev, pn);
  ed_res f, r;
  ed_align(prefix, pn, cs, L, ED_HW, 0, &f);
  ed_align(prev, pn, cs, L, ED_HW, 0, &r);
  free(prev);
  if (f.ed > r.ed) dor_reverse_complement(cs, L);
  ed_res cp, csuf;
  ed_align(prefix, pn, cs, L, ED_HW, 2, &cp);
  uint32_t csStart = infix_start(&cp);
  free(cp.aln);
  ed_align(suffix, sn, cs, L, ED_HW, 2, &csuf);
  uint32_t csEnd = (uint32_t)csuf.end_loc;
  free(csuf.aln);
  if (csStart < csEnd && csEnd < (uint32_t)L) {
    memmove(cs, cs + csStart, (size_t)(csEnd - csStart));
    L = (int)(csEnd - csStart);
  }
  return L;
}

static int msa_wfa_core(const dellyhip_params* p, int n, const char* blob, const uint64_t* off, const char* prefix, int pn,
                        const char* suffix, int sn, char** cs_out, int* cs_len) {
  const unsigned char* eq = iupac_equalities();
  uint32_t* hitI = (uint32_t*)malloc(sizeof(uint32_t) * DOR_KTAB);
  uint32_t* hitJ = (uint32_t*)malloc(sizeof(uint32_t) * DOR_KTAB);
  int32_t* edit = (int32_t*)calloc((size_t)n * n, sizeof(int32_t));
#define RD(k) (blob + off[k])
#define RL(k) ((uint32_t)(off[(k) + 1] - off[k]))
  for (int i = 0; i < n; ++i) {
    uint32_t lenI = RL(i);
    fill_kmer_table(RD(i), lenI, hitI);
    for (int j = i + 1; j < n; ++j) {
      uint32_t lenJ = RL(j);
      fill_kmer_table(RD(j), lenJ, hitJ);
      int32_t bd = best_diagonal(hitI, hitJ, lenI, lenJ);
      const char *sI, *sJ;
      uint32_t seqlen;
      if (bd >= 0) { seqlen = (lenI - (uint32_t)bd < lenJ) ? lenI - (uint32_t)bd : lenJ; sI = RD(i) + bd; sJ = RD(j); }
      else { seqlen = (lenJ + (uint32_t)bd < lenI) ? lenJ + (uint32_t)bd : lenI; sI = RD(i); sJ = RD(j) + (-bd); }
      /* std::string::substr clamps the length to what is left (and throws past the end: not mirrored) */
      uint32_t lI = seqlen, lJ = seqlen;
      { uint32_t oI = (bd >= 0) ? (uint32_t)bd : 0, oJ = (bd >= 0) ? 0 : (uint32_t)(-bd);
        if (lI > lenI - oI) lI = lenI - oI;
        if (lJ > lenJ - oJ) lJ = lenJ - oJ; }
      ed_res a;
      ed_align(sI, (int)lI, sJ, (int)lJ, ED_NW, 0, &a);
      int32_t mx = (int32_t)(lI > lJ ? lI : lJ);
      int32_t score = (a.ed * 1000) / mx;
      edit[i * n + j] = edit[j * n + i] = score;
    }
  }
  uint32_t bestIdx = 0;
  int32_t bestVal = (int32_t)RL(0);
  int* dist = (int*)malloc(sizeof(int) * (size_t)n);
  for (int i = 0; i < n; ++i) {
    for (int j = 0; j < n; ++j) dist[j] = edit[i * n + j];
    qsort(dist, (size_t)n, sizeof(int), cmp_int);
    if (dist[n / 2] < bestVal) { bestVal = dist[n / 2]; bestIdx = (uint32_t)i; }
  }
  free(dist);
  int32_t* qs = (int32_t*)malloc(sizeof(int32_t) * 2 * (size_t)n);
  int nq = 1;
  qs[0] = 0; qs[1] = (int32_t)bestIdx;
  for (int j = 0; j < n; ++j)
    if ((uint32_t)j != bestIdx) { qs[2 * nq] = edit[bestIdx * n + j]; qs[2 * nq + 1] = j; ++nq; }
  qsort(qs, (size_t)nq, 2 * sizeof(int32_t), cmp_pair);
  uint32_t lastIdx = (uint32_t)(0.8 * nq);
  if (lastIdx < 3) lastIdx = 3;
  int nsel = 0;
  int* sel = (int*)malloc(sizeof(int) * (size_t)n);
  for (uint32_t i = 0; i < (uint32_t)nq && i < lastIdx; ++i) sel[nsel++] = qs[2 * i + 1];
  free(qs);
  free(edit);
  /* superstring (assemble.h:598-660) */
  int sl = (int)RL(sel[0]);
  char* super = (char*)malloc((size_t)sl + 1);
  memcpy(super, RD(sel[0]), (size_t)sl);
  for (int i = 1; i < nsel; ++i) {
    uint32_t lenI = (uint32_t)sl;
    fill_kmer_table(super, lenI, hitI);
    uint32_t lenJ = RL(sel[i]);
    fill_kmer_table(RD(sel[i]), lenJ, hitJ);
    int32_t bd = best_diagonal(hitI, hitJ, lenI, lenJ);
    uint32_t preI = 0, postI = 0, preJ = 0, postJ = 0, seqlen = 0;
    if (bd >= 0) {
      seqlen = (lenI - (uint32_t)bd < lenJ) ? lenI - (uint32_t)bd : lenJ;
      preI = (uint32_t)bd; postI = lenI - ((uint32_t)bd + seqlen); preJ = 0; postJ = lenJ - seqlen;
    } else {
      seqlen = (lenJ + (uint32_t)bd < lenI) ? lenJ + (uint32_t)bd : lenI;
      preI = 0; postI = lenI - seqlen; preJ = (uint32_t)(-bd); postJ = lenJ - ((uint32_t)(-bd) + seqlen);
    }
    if (preI > preJ && postI > postJ) {
      /* nested */
    } else if (preJ > preI && postJ > postI) {
      free(super);
      sl = (int)lenJ;
      super = (char*)malloc((size_t)sl + 1);
      memcpy(super, RD(sel[i]), (size_t)sl);
    } else {
      const char* sI = (bd >= 0) ? super + bd : super;
      const char* sJ = (bd >= 0) ? RD(sel[i]) : RD(sel[i]) + (-bd);
      ed_res cg;
      ed_align(sI, (int)seqlen, sJ, (int)seqlen, ED_NW, 2, &cg);
      int on = 0;
      char* outStr = build_superstring(super, RD(sel[i]), &cg, preI, postI, preJ, postJ, (size_t)lenI + lenJ + 8, &on);
      free(cg.aln);
      free(super);
      super = outStr;
      sl = on;
    }
  }
  free(hitI);
  free(hitJ);
  /* progressive HW alignment of every selected read to the running consensus (:662-686) */
  amat al = amat_new(1, sl);
  memcpy(al.d, super, (size_t)sl);
  free(super);
  for (int i = 0; i < nsel; ++i) {
    char* astr = (char*)malloc((size_t)al.cols + 1);
    consensus_wfa(&al, astr);
    ed_res c;
    ed_align_eq(RD(sel[i]), (int)RL(sel[i]), astr, al.cols, ED_HW, 2, eq, &c);
    amat nx = convert_alignment_hw(RD(sel[i]), &al, &c);
    free(c.aln);
    free(astr);
    amat_free(&al);
    al = nx;
  }
  char* cs = (char*)malloc((size_t)al.cols + 1);
  int L = consensus_core(p, &al, cs);
  amat_free(&al);
  if (pn > 0 && sn > 0) L = trim_consensus(prefix, pn, suffix, sn, cs, L);
  else {
    int32_t trim = (int32_t)(0.05 * L);
    if (trim > 50) trim = 50;
    int32_t len = L - 2 * trim;
    if (len > 100) { memmove(cs, cs + trim, (size_t)len); L = len; }
  }
  free(sel);
#undef RD
#undef RL
  *cs_out = cs;
  *cs_len = L;
  return nsel;
}

int dor_msa_wfa(const dellyhip_params* p, int n_reads, const char* blob, const uint64_t* off, const char* prefix, int pn,
                const char* suffix, int sn, char* cs, int cap, int* cs_len) {
  char* c = NULL;
  int L = 0;
  int rows = msa_wfa_core(p, n_reads, blob, off, prefix, pn, suffix, sn, &c, &L);
  *cs_len = L;
  if (L <= cap) memcpy(cs, c, (size_t)L);
  free(c);
  return rows;
}

/* ------------------------------------------------------------------------ */
/* batch driver: loop body of src/shortpe.h:183-197 (msa + alignConsensus)    */

typedef struct {
  const dellyhip_params* p;
  int n_chr;
  const char* const* chr_seq;
  const int64_t* chr_len;
  int n_junc;
  const dellyhip_junction* junc;
  const char* blob;
  const uint64_t* off;
  dellyhip_result* results;
  char* out_blob;
  uint64_t out_cap;
  _Atomic uint64_t used;
  _Atomic uint32_t next;
  int with_msa, want_alignment;
  dellyhip_probes* probes;   /* non-NULL: _generateProbes flavour (dor_generate_probes) */
  _Atomic uint64_t probe_used;
  char* probe_blob;
  uint64_t probe_cap;
} batch_t;

static uint64_t blob_put(batch_t* b, const char* p, uint64_t n) {
  uint64_t o = atomic_fetch_add(&b->used, n);
  if (b->out_blob == NULL || o + n > b->out_cap) return UINT64_MAX;
  if (n) memcpy(b->out_blob + o, p, n);
  return o;
}

/* _cutRefStart / _cutRefEnd  src/coverage.h:117-162 */
static int cut_ref(int rStart, int rEnd, int offset, int bpPoint, int svt, int is_end) {
  const int ct = is_tra(svt) ? svt - 5 : svt;   /* _getSpanOrientation src/tags.h:33-40; plain SVs test svt == 3 */
  int anchor;
  if (ct == 3) anchor = (!bpPoint) ? rEnd : rStart;
  else anchor = bpPoint ? rEnd : rStart;
  return is_end ? anchor + offset : anchor - offset;
}

static uint64_t probe_put(batch_t* b, const char* p, uint64_t n) {
  uint64_t o = atomic_fetch_add(&b->probe_used, n);
  if (b->probe_blob && o + n <= b->probe_cap) memcpy(b->probe_blob + o, p, (size_t)n);
  return o;
}

/* the loop over bpPoint of _generateProbes, src/coverage.h:230-258 (substr semantics: the start lies inside the
 * string whenever _findSplit succeeded, the length is clipped at its end) */
static void probes_cut(batch_t* b, const dellyhip_junction* J, const adesc* ad, const char* cons, int m, const char* ref, int n,
                       dellyhip_probes* O) {
  const dellyhip_params* c = b->p;
  const int mfs = c->minimum_flank_size;
  O->ok = 1;
  O->hom_left = ad->homLeft;
  O->hom_right = ad->homRight;
  for (int bp = 0; bp < 2; ++bp) {
    const int anchor = bp ? ad->cEnd : ad->cStart;
    const int cs = anchor - ad->homLeft - mfs, ce = anchor + ad->homRight + mfs;
    const int rs = cut_ref(ad->rStart, ad->rEnd, ad->homLeft + mfs, bp, J->svt, 0);
    const int re = cut_ref(ad->rStart, ad->rEnd, ad->homRight + mfs, bp, J->svt, 1);
    int cl = ce - cs, rl = re - rs;
    if (cl > m - cs) cl = m - cs;
    if (rl > n - rs) rl = n - rs;
    O->cons_len[bp] = cl;
    O->ref_len[bp] = rl;
    O->cons_off[bp] = probe_put(b, cons + cs, (uint64_t)cl);
    O->ref_off[bp] = probe_put(b, ref + rs, (uint64_t)rl);
  }
}

static void refine_one(batch_t* b, const dellyhip_junction* J, dellyhip_result* R) {
  const dellyhip_params* c = b->p;
  memset(R, 0, sizeof(*R));
  R->svid = J->svid;
  R->sv_start = J->sv_start;
  R->sv_end = J->sv_end;
  R->ins_len = J->ins_len;
  R->score_unsplit = R->score_best = R->cons_left = R->ref_left = R->ref_right = -1;
  R->matches = R->mismatches = -1;

  char* cons = NULL;
  int m = 0;
  if (b->with_msa) {
    if (b->with_msa != 2 && J->n_seq <= 1) return; /* shortpe.h:166-171 */
    if (b->with_msa == 2) { /* long-read loop body: src/assemble.h:836-861 (the small-inversion
                             * substring of :842-854 is not mirrored) */
      if (J->n_seq < 1) return;
      if (J->svt == 4) {
        const char* seq = b->chr_seq[J->chr];
        const int32_t seqlen = (int32_t)(uint32_t)b->chr_len[J->chr];
        const int32_t p0 = imax(J->sv_start - c->min_cons_window, 0), p1 = J->sv_start;
        const int32_t s0 = J->sv_start, s1 = imin(seqlen, J->sv_start + c->min_cons_window);
        sbuf pre, suf;
        sb_init(&pre); sb_init(&suf);
        sb_upper(&pre, seq, p0, p1);
        sb_upper(&suf, seq, s0, s1);
        R->sr_support = msa_wfa_core(c, J->n_seq, b->blob, b->off + J->seq_first, pre.d, (int)pre.n, suf.d, (int)suf.n, &cons, &m);
        free(pre.d); free(suf.d);
      } else
      R->sr_support = msa_edlib_core(c, J->n_seq, b->blob, b->off + J->seq_first, &cons, &m);
    } else
    R->sr_support = msa_core(c, J->n_seq, b->blob, b->off + J->seq_first, &cons, &m);
    /* NOTE: off + seq_first keeps absolute offsets into blob */
  } else {
    m = (int)(b->off[J->seq_first + 1] - b->off[J->seq_first]);
    cons = (char*)malloc((size_t)m + 1);
    memcpy(cons, b->blob + b->off[J->seq_first], (size_t)m);
  }
  R->cons_len = m;
  R->cons_off = blob_put(b, cons, (uint64_t)m);
  /* src/assemble.h:840-853: small inversions in the long-read loop -- only the middle svSize letters are
   * aligned; the consensus is restored afterwards and consBp shifted */
  char* cons_full = NULL;
  int32_t off_small = 0;
  if (b->with_msa == 2 && J->svt != 4) {
    int32_t svSize = J->sv_end - J->sv_start;
    if ((J->svt == 0 || J->svt == 1) && svSize < m) {
      off_small = (int32_t)(((size_t)m - (size_t)svSize) / 2);
      cons_full = cons;
      size_t take = (svSize > 0) ? (size_t)svSize : 0;
      if ((size_t)off_small > (size_t)m) off_small = m;           /* std::string::substr semantics */
      if (take > (size_t)m - (size_t)off_small) take = (size_t)m - (size_t)off_small;
      cons = (char*)malloc(take + 1);
      memcpy(cons, cons_full + off_small, take);
      m = (int)take;
    }
  }

  /* alignConsensus  split.h:644-666 (bit 1 of reserved: _generateProbes, src/coverage.h:196-217, has no length test) */
  if (!(c->reserved & 2) && m < (2 * c->minimum_flank_size + J->ins_len)) {
    free(cons);
    free(cons_full);
    return;
  }
  bpoint bp;
  bp.svStartBeg = bp.svStartEnd = bp.svStart = J->sv_start;
  bp.svEndBeg = bp.svEndEnd = bp.svEnd = J->sv_end;
  bp.svt = J->svt;
  bp.chr = J->chr;
  bp.chr2 = J->chr2;
  if (J->svt == 4) { /* split.h:650-652: size_t arithmetic, then (int32_t) */
    int32_t bufferSpace = imax((int32_t)(((size_t)m - (size_t)J->ins_len) / 3), c->minimum_flank_size);
    init_breakpoint(b->chr_len, &bp, bufferSpace, J->svt);
  } else init_breakpoint(b->chr_len, &bp, m, J->svt);
  sbuf part1, ref;
  sb_init(&part1);
  sb_init(&ref);
  if (bp.chr != bp.chr2) get_sv_ref(c, b->chr_seq[J->chr2], &bp, bp.chr2, J->svt, &part1, &part1);
  /* (part1 is empty on entry, so aliasing in/out is harmless: the chr2 branch never reads it) */
  get_sv_ref(c, b->chr_seq[J->chr], &bp, bp.chr, J->svt, &part1, &ref);
  int n = (int)ref.n;
  R->ref_len = n;

  /* _alignConsensus split.h:560-642; realign (split.h:564-572) = bit 0 of params.reserved */
  if ((c->reserved & 1) && !(b->with_msa == 2 && J->svt == 4)) { /* assemble.h:859: insertions pass realign = false */
    char* revc = (char*)malloc((size_t)m + 1);
    memcpy(revc, cons, (size_t)m);
    dor_reverse_complement(revc, m);
    ed_res f, r;
    ed_align(ref.d, n, cons, m, ED_NW, 0, &f);
    ed_align(ref.d, n, revc, m, ED_NW, 0, &r);
    if (r.ed < f.ed) {
      memcpy(cons, revc, (size_t)m);
      if (!cons_full && b->out_blob && R->cons_off != UINT64_MAX) memcpy(b->out_blob + R->cons_off, cons, (size_t)m); /* sv.consensus = revc */
    }
    free(revc);
  }
  amat al;
  int diag[5];
  int found;
  if (J->svt == 4) { /* _consRefAlignment split.h:546-552; diag = {csStart, csEnd, bestJoin, leftEnd, rightStart} */
    found = split_align(cons, m, ref.d, n, &al, diag);
    if (found < 0) {
      R->status = DELLYHIP_E_LIMIT;
      found = 0;
    }
  } else found = long_needle_core(cons, m, ref.d, n, &al, diag);
  R->score_unsplit = diag[0];
  R->score_best = diag[1];
  R->cons_left = diag[2];
  R->ref_left = diag[3];
  R->ref_right = diag[4];
  if (found) {
    if (b->want_alignment) {
      R->aln_off = blob_put(b, al.d, (uint64_t)2 * (uint64_t)al.cols);
      R->aln_len = al.cols;
    }
    adesc ad;
    memset(&ad, 0, sizeof(ad));
    if (find_split(c, cons, m, ref.d, n, &al, &ad, J->svt)) {
      R->c_start = ad.cStart; R->c_end = ad.cEnd; R->r_start = ad.rStart; R->r_end = ad.rEnd;
      R->hom_left = ad.homLeft; R->hom_right = ad.homRight;
      R->matches = (int32_t)ad.ma;
      R->mismatches = (int32_t)ad.mm;
      if (b->probes) probes_cut(b, J, &ad, cons, m, ref.d, n, &b->probes[J - b->junc]);
      uint32_t gs = 0, ge = 0;
      if (coord_transform(c, (uint64_t)n, &bp, &ad, &gs, &ge, J->svt) && (is_tra(J->svt) || gs < ge)) {
        /* exact alleles split.h:606-624 */
        if (J->sv_end - J->sv_start <= c->indelsize && (J->svt == 2 || J->svt == 4)) {
          char* refV = (char*)malloc((size_t)al.cols + 2);
          char* altV = (char*)malloc((size_t)al.cols + 2);
          int nr = 0, na = 0, cpos = 0, inSV = 0;
          for (int j = 0; j < al.cols; ++j) {
            if (AT(al, 0, j) != '-') {
              ++cpos;
              if (cpos == ad.cStart) inSV = 1;
              else if (cpos == ad.cEnd) inSV = 0;
            }
            if (inSV) {
              if (AT(al, 0, j) != '-') altV[na++] = AT(al, 0, j);
              if (AT(al, 1, j) != '-') refV[nr++] = AT(al, 1, j);
            }
          }
          char* both = (char*)malloc((size_t)nr + na + 2);
          memcpy(both, refV, (size_t)nr);
          both[nr] = ',';
          memcpy(both + nr + 1, altV, (size_t)na);
          R->allele_off = blob_put(b, both, (uint64_t)nr + na + 1);
          R->allele_len = nr + na + 1;
          free(refV);
          free(altV);
          free(both);
        }
        R->ok = 1;
        R->sv_start = (int32_t)gs;
        R->sv_end = (int32_t)ge;
        R->sr_align_quality = ad.percId;
        R->ins_len = ad.cEnd - ad.cStart - 1;
        R->cons_bp = ad.cStart + off_small; /* (+ offsetTmpCons, src/assemble.h:852) */
        R->hom_len = imax(0, ad.homLeft + ad.homRight - 2);
        R->ci_wiggle = imax(ad.homLeft, ad.homRight);
      }
    }
    amat_free(&al);
  }
  free(part1.d);
  free(ref.d);
  free(cons);
  free(cons_full);
}

static void* worker(void* arg) {
  batch_t* b = (batch_t*)arg;
  for (;;) {
    uint32_t i = atomic_fetch_add(&b->next, 1);
    if (i >= (uint32_t)b->n_junc) break;
    refine_one(b, &b->junc[i], &b->results[i]);
  }
  return NULL;
}

/* CPU baseline timer (bench.py cpu_baseline, kind "port"): the same worker loop, `reps` passes over the batch,
 * results discarded (one scratch record per thread), clock around thread start .. join inside this function. */
typedef struct { batch_t* b; uint64_t total; _Atomic uint64_t* next; _Atomic int64_t* oks; } timed_t;
static void* timed_worker(void* arg) {
  timed_t* t = (timed_t*)arg;
  dellyhip_result R;
  int64_t mine = 0;
  for (;;) {
    uint64_t i = atomic_fetch_add(t->next, 1);
    if (i >= t->total) break;
    refine_one(t->b, &t->b->junc[i % (uint64_t)t->b->n_junc], &R);
    mine += R.ok;
  }
  atomic_fetch_add(t->oks, mine);
  return NULL;
}
int dor_time_refine_batch(const dellyhip_params* p, int n_chr, const char* const* chr_seq,
                          const int64_t* chr_len, int n_junc, const dellyhip_junction* junc,
                          const char* blob, const uint64_t* off, int with_msa, int n_threads, int reps,
                          double* seconds, int64_t* n_ok) {
  batch_t b;
  b.p = p; b.n_chr = n_chr; b.chr_seq = chr_seq; b.chr_len = chr_len; b.n_junc = n_junc;
  b.junc = junc; b.blob = blob; b.off = off; b.results = NULL; b.out_blob = NULL;
  b.out_cap = 0; b.with_msa = with_msa; b.want_alignment = 0;
  b.probes = NULL; b.probe_blob = NULL; b.probe_cap = 0;
  atomic_init(&b.probe_used, 0);
  atomic_init(&b.used, 0);
  atomic_init(&b.next, 0);
  _Atomic uint64_t next;
  _Atomic int64_t oks;
  atomic_init(&next, 0);
  atomic_init(&oks, 0);
  timed_t t = {&b, (uint64_t)(reps > 1 ? reps : 1) * (uint64_t)(n_junc > 0 ? n_junc : 0), &next, &oks};
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  if (n_threads <= 1) timed_worker(&t);
  else {
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)n_threads);
    for (int k = 0; k < n_threads; ++k) pthread_create(&th[k], NULL, timed_worker, &t);
    for (int k = 0; k < n_threads; ++k) pthread_join(th[k], NULL);
    free(th);
  }
  clock_gettime(CLOCK_MONOTONIC, &t1);
  if (seconds) *seconds = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
  if (n_ok) *n_ok = atomic_load(&oks);
  return 0;
}

int dor_refine_batch(const dellyhip_params* p, int n_chr, const char* const* chr_seq,
                     const int64_t* chr_len, int n_junc, const dellyhip_junction* junc,
                     const char* blob, const uint64_t* off, dellyhip_result* results,
                     char* out_blob, uint64_t out_cap, uint64_t* out_used, int with_msa,
                     int want_alignment, int n_threads) {
  batch_t b;
  b.p = p; b.n_chr = n_chr; b.chr_seq = chr_seq; b.chr_len = chr_len; b.n_junc = n_junc;
  b.junc = junc; b.blob = blob; b.off = off; b.results = results; b.out_blob = out_blob;
  b.out_cap = out_cap; b.with_msa = with_msa; b.want_alignment = want_alignment;
  b.probes = NULL; b.probe_blob = NULL; b.probe_cap = 0;
  atomic_init(&b.probe_used, 0);
  atomic_init(&b.used, 0);
  atomic_init(&b.next, 0);
  if (n_threads <= 1) worker(&b);
  else {
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)n_threads);
    for (int t = 0; t < n_threads; ++t) pthread_create(&th[t], NULL, worker, &b);
    for (int t = 0; t < n_threads; ++t) pthread_join(th[t], NULL);
    free(th);
  }
  uint64_t used = atomic_load(&b.used);
  if (out_used) *out_used = used;
  return (out_blob && used > out_cap) ? -1 : 0;
}


/* ------------------------------------------------------------------------------------------------
 * Split-read genotyping classifier: the worker body of process_batch,
 * src/coverage.h:418-434, on top of _editDistanceHW (:107-115).
 * edlib's HW distance with threshold k (src/edlib.cpp:157-170 empty operands; :545-700: the best
 * last-row value over all target columns, k = min(queryLength, k), -1 when it exceeds k) is restated
 * as the plain unit-cost DP with a free first row -- the distance is unique.
 * ------------------------------------------------------------------------------------------------ */
static int hw_distance_k(const char* q, int qn, const char* t, int tn, int k) {
  if (qn == 0 || tn == 0) return qn;
  int* col = (int*)malloc(sizeof(int) * (size_t)(qn + 1));
  for (int i = 0; i <= qn; ++i) col[i] = i;
  int best = qn;
  for (int j = 1; j <= tn; ++j) {
    int diag = col[0];   /* D[0][j-1] = 0 */
    col[0] = 0;
    for (int i = 1; i <= qn; ++i) {
      const int up_left = diag + (q[i - 1] != t[j - 1]);
      diag = col[i];
      int v = up_left;
      if (col[i] + 1 < v) v = col[i] + 1;         /* D[i][j-1] + 1 */
      if (col[i - 1] + 1 < v) v = col[i - 1] + 1; /* D[i-1][j] + 1 */
      col[i] = v;
    }
    if (col[qn] < best) best = col[qn];
  }
  free(col);
  if (k < 0) return best;          /* edlib: k < 0 = search until found */
  if (qn < k) k = qn;              /* src/edlib.cpp:563-565 */
  return best <= k ? best : -1;
}

/* _editDistanceHW  src/coverage.h:107-115 */
static double edit_distance_hw_score(float flank_quality, const char* q, int qn, const char* t, int tn, int* dist) {
  double score = 0;
  const int k = (int)(2 * flank_quality * (size_t)qn);   /* float product, truncated by edlibNewAlignConfig(int k, ...) */
  const int d = hw_distance_k(q, qn, t, tn, k);
  if (d != -1) score = ((1.0 - flank_quality) * (double)qn) / (double)(d + 1);
  *dist = d;
  return score;
}

typedef struct {
  const dellyhip_params* p; const dellyhip_align_job* jobs; const char* blob; dellyhip_align_result* out;
  uint64_t n; volatile uint64_t* next;
} cls_work;

static void* cls_worker(void* arg) {
  cls_work* w = (cls_work*)arg;
  for (;;) {
    const uint64_t i = __sync_fetch_and_add(w->next, 1);
    if (i >= w->n) break;
    const dellyhip_align_job* J = &w->jobs[i];
    dellyhip_align_result r;
    memset(&r, 0, sizeof r);
    r.type = 'N';
    const char* seq = w->blob + J->seq_off;
    const double scoreAlt = edit_distance_hw_score(w->p->flank_quality, w->blob + J->cons_off, (int)J->cons_len, seq, (int)J->seq_len, &r.dist_alt);
    const double scoreRef = edit_distance_hw_score(w->p->flank_quality, w->blob + J->ref_off, (int)J->ref_len, seq, (int)J->seq_len, &r.dist_ref);
    if ((scoreRef > 0.7) || (scoreAlt > 0.7)) {   /* src/coverage.h:424-433 */
      r.sv_id = J->sv_id;
      r.file_index = J->file_index;
      if (scoreRef > scoreAlt) {
        int qv = (int)(scoreRef * 35); if ((int)J->qual < qv) qv = (int)J->qual; if (qv > 255) qv = 255;
        r.type = 'R'; r.qual = (uint8_t)qv;
      } else {
        int qv = (int)(scoreAlt * 35); if ((int)J->qual < qv) qv = (int)J->qual; if (qv > 255) qv = 255;
        r.type = 'A'; r.qual = (uint8_t)qv;
      }
    }
    w->out[i] = r;
  }
  return NULL;
}

static double now_seconds(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

int dor_classify_reads(const dellyhip_params* p, uint64_t n_jobs, const dellyhip_align_job* jobs, const char* blob,
                       dellyhip_align_result* out, int n_threads, int with_dist, double* worker_seconds) {
  (void)with_dist;   /* the restatement computes each distance once and always reports it */
  volatile uint64_t next = 0;
  cls_work w = {p, jobs, blob, out, n_jobs, &next};
  const double t0 = now_seconds();
  if (n_threads <= 1) cls_worker(&w);
  else {
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)n_threads);
    for (int t = 0; t < n_threads; ++t) pthread_create(&th[t], NULL, cls_worker, &w);
    for (int t = 0; t < n_threads; ++t) pthread_join(th[t], NULL);
    free(th);
  }
  if (worker_seconds) *worker_seconds = now_seconds() - t0;
  return 0;
}


/* ------------------------------------------------------------------------------------------------
 * Long-read genotyping: _editDistanceNW, src/genotype.h:21-30 -- edlib's global
 * edit distance with k = -1 (always found), restated as the two-row unit-cost DP.
 * ------------------------------------------------------------------------------------------------ */
static int nw_distance(const char* q, int qn, const char* t, int tn) {
  if (qn == 0 || tn == 0) return imax(qn, tn);   /* src/edlib.cpp:157-163 */
  int* col = (int*)malloc(sizeof(int) * (size_t)(qn + 1));
  for (int i = 0; i <= qn; ++i) col[i] = i;
  for (int j = 1; j <= tn; ++j) {
    int diag = col[0];
    col[0] = j;
    for (int i = 1; i <= qn; ++i) {
      int v = diag + (q[i - 1] != t[j - 1]);
      diag = col[i];
      if (col[i] + 1 < v) v = col[i] + 1;
      if (col[i - 1] + 1 < v) v = col[i - 1] + 1;
      col[i] = v;
    }
  }
  const int d = col[qn];
  free(col);
  return d;
}

typedef struct { const dellyhip_nw_job* jobs; const char* blob; int32_t* out; uint64_t n; volatile uint64_t* next; } nw_work;

static void* nw_worker(void* arg) {
  nw_work* w = (nw_work*)arg;
  for (;;) {
    const uint64_t i = __sync_fetch_and_add(w->next, 1);
    if (i >= w->n) break;
    const dellyhip_nw_job* J = &w->jobs[i];
    w->out[i] = nw_distance(w->blob + J->query_off, (int)J->query_len, w->blob + J->target_off, (int)J->target_len);
  }
  return NULL;
}

int dor_edit_distance_nw_batch(uint64_t n_jobs, const dellyhip_nw_job* jobs, const char* blob, int32_t* out, int n_threads,
                               double* worker_seconds) {
  volatile uint64_t next = 0;
  nw_work w = {jobs, blob, out, n_jobs, &next};
  const double t0 = now_seconds();
  if (n_threads <= 1) nw_worker(&w);
  else {
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)n_threads);
    for (int t = 0; t < n_threads; ++t) pthread_create(&th[t], NULL, nw_worker, &w);
    for (int t = 0; t < n_threads; ++t) pthread_join(th[t], NULL);
    free(th);
  }
  if (worker_seconds) *worker_seconds = now_seconds() - t0;
  return 0;
}


/* ------------------------------------------------------------------------------------------------
 * Probe generation: the per-SV body of _generateProbes, src/coverage.h:196-258 --
 * window, _consRefAlignment, _findSplit (all shared with alignConsensus above, without its length
 * test), then the probe substrings and the BpRegion fields.  Single-threaded: the reference loop is.
 * ------------------------------------------------------------------------------------------------ */
int dor_generate_probes(const dellyhip_params* p, int n_chr, const char* const* chr_seq, const int64_t* chr_len, int n_junc,
                        const dellyhip_junction* junc, const char* blob, const uint64_t* off, dellyhip_probes* probes,
                        char* out_blob, uint64_t out_cap, uint64_t* out_used) {
  dellyhip_params pp = *p;
  pp.reserved |= 2;
  dellyhip_result* results = (dellyhip_result*)calloc((size_t)(n_junc > 0 ? n_junc : 1), sizeof(dellyhip_result));
  batch_t b;
  b.p = &pp; b.n_chr = n_chr; b.chr_seq = chr_seq; b.chr_len = chr_len; b.n_junc = n_junc;
  b.junc = junc; b.blob = blob; b.off = off; b.results = results; b.out_blob = NULL;
  b.out_cap = 0; b.with_msa = 0; b.want_alignment = 0;
  b.probes = probes; b.probe_blob = out_blob; b.probe_cap = out_cap;
  atomic_init(&b.probe_used, 0);
  atomic_init(&b.used, 0);
  atomic_init(&b.next, 0);
  const int mfs = pp.minimum_flank_size;
  for (int i = 0; i < n_junc; ++i) {
    dellyhip_probes* O = &probes[i];
    memset(O, 0, sizeof *O);
    O->svid = junc[i].svid;
    O->region_start[0] = imax(0, junc[i].sv_start - mfs);                          /* src/coverage.h:244-245 */
    O->region_end[0] = (int32_t)(((uint32_t)(junc[i].sv_start + mfs) < (uint32_t)chr_len[junc[i].chr]) ? (uint32_t)(junc[i].sv_start + mfs) : (uint32_t)chr_len[junc[i].chr]);
    O->bppos[0] = junc[i].sv_start;
    O->region_start[1] = imax(0, junc[i].sv_end - mfs);                            /* :235-236 */
    O->region_end[1] = (int32_t)(((uint32_t)(junc[i].sv_end + mfs) < (uint32_t)chr_len[junc[i].chr2]) ? (uint32_t)(junc[i].sv_end + mfs) : (uint32_t)chr_len[junc[i].chr2]);
    O->bppos[1] = junc[i].sv_end;
  }
  worker(&b);
  for (int i = 0; i < n_junc; ++i) probes[i].status = results[i].status;
  free(results);
  uint64_t used = atomic_load(&b.probe_used);
  if (out_used) *out_used = used;
  return (out_blob && used > out_cap) ? -1 : 0;
}
