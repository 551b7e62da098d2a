/* ossl_check.c -- independent cross-check and competitor baseline through OpenSSL's libcrypto.
 *
 * TEST INFRASTRUCTURE ONLY (like everything under oracle/): nothing in the product links or loads this.
 *
 * Two purposes:
 *  - an implementation of P-256 / secp256k1 point multiplication that shares no code and no algorithm with
 *    either aguinet/ecsimd or this repository's restatement of it, so a level-A (affine) agreement of the HIP
 *    path with it is evidence independent of the oracle;
 *  - the restatement of the reference's competitor benchmark, benchs/p256_ref.cpp:55-91 (bench_openssl:
 *    EC_POINT_mul(curve, P, NULL, randp, prv, ctx) in a loop), here over a batch and over several threads.
 *
 * Values cross the interface as 4 x u64 little-endian limbs (the C ABI's element layout), classical domain.
 */
#include <openssl/bn.h>
#include <openssl/ec.h>
#include <openssl/ecdsa.h>
#include <openssl/obj_mac.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define API __attribute__((visibility("default")))

static int nid_of(int curve) { return curve == 0 ? NID_X9_62_prime256v1 : curve == 1 ? NID_secp256k1 : -1; }

typedef struct {
  int curve, mode;                       /* mode 0: k*P   1: k*G   2: u1*G + u2*Q */
  const uint64_t *k, *k2, *x, *y;
  uint64_t *ox, *oy;
  uint8_t* inf;
  size_t begin, end;
  int rc;
} job_t;

static void* worker(void* arg) {
  job_t* j = (job_t*)arg;
  j->rc = -1;
  EC_GROUP* g = EC_GROUP_new_by_curve_name(nid_of(j->curve));
  BN_CTX* ctx = BN_CTX_new();
  BIGNUM *k = BN_new(), *k2 = BN_new(), *x = BN_new(), *y = BN_new();
  EC_POINT *P = g ? EC_POINT_new(g) : NULL, *R = g ? EC_POINT_new(g) : NULL;
  if (!g || !ctx || !k || !k2 || !x || !y || !P || !R) goto done;
  for (size_t i = j->begin; i < j->end; ++i) {
    if (!BN_lebin2bn((const unsigned char*)(j->k + 4 * i), 32, k)) goto done;
    if (j->mode != 1) {
      if (!BN_lebin2bn((const unsigned char*)(j->x + 4 * i), 32, x)) goto done;
      if (!BN_lebin2bn((const unsigned char*)(j->y + 4 * i), 32, y)) goto done;
      if (!EC_POINT_set_affine_coordinates(g, P, x, y, ctx)) goto done;      /* rejects points off the curve */
    }
    int ok;
    if (j->mode == 0) ok = EC_POINT_mul(g, R, NULL, P, k, ctx);                 /* p256_ref.cpp:84 */
    else if (j->mode == 1) ok = EC_POINT_mul(g, R, k, NULL, NULL, ctx);         /* p256_ref.cpp:77 */
    else {
      if (!BN_lebin2bn((const unsigned char*)(j->k2 + 4 * i), 32, k2)) goto done;
      ok = EC_POINT_mul(g, R, k, P, k2, ctx);                                   /* k*G + k2*P */
    }
    if (!ok) goto done;
    if (j->ox) {
      if (EC_POINT_is_at_infinity(g, R)) {
        j->inf[i] = 1;
        memset(j->ox + 4 * i, 0, 32); memset(j->oy + 4 * i, 0, 32);
      } else {
        j->inf[i] = 0;
        if (!EC_POINT_get_affine_coordinates(g, R, x, y, ctx)) goto done;
        if (BN_bn2lebinpad(x, (unsigned char*)(j->ox + 4 * i), 32) != 32) goto done;
        if (BN_bn2lebinpad(y, (unsigned char*)(j->oy + 4 * i), 32) != 32) goto done;
      }
    }
  }
  j->rc = 0;
done:
  EC_POINT_free(P); EC_POINT_free(R);
  BN_free(k); BN_free(k2); BN_free(x); BN_free(y);
  BN_CTX_free(ctx); EC_GROUP_free(g);
  return NULL;
}

static int run(int curve, int mode, const uint64_t* k, const uint64_t* k2, const uint64_t* x, const uint64_t* y,
               uint64_t* ox, uint64_t* oy, uint8_t* inf, size_t n, int threads) {
  if (nid_of(curve) < 0) return -2;
  if (threads < 1) threads = 1;
  if (threads > 256) threads = 256;
  if ((size_t)threads > n) threads = n ? (int)n : 1;
  job_t jobs[256];
  pthread_t tid[256];
  for (int t = 0; t < threads; ++t) {
    job_t j = {curve, mode, k, k2, x, y, ox, oy, inf, n * t / threads, n * (t + 1) / threads, 0};
    jobs[t] = j;
    if (threads == 1) worker(&jobs[t]);
    else if (pthread_create(&tid[t], NULL, worker, &jobs[t])) return -3;
  }
  int rc = 0;
  for (int t = 0; t < threads; ++t) {
    if (threads > 1) pthread_join(tid[t], NULL);
    if (jobs[t].rc) rc = jobs[t].rc;
  }
  return rc;
}

/* (ox, oy) = k * (x, y), affine classical; inf[i] = 1 where the result is the point at infinity. */
API int ossl_scalar_mult(int curve, const uint64_t* k, const uint64_t* x, const uint64_t* y, uint64_t* ox,
                         uint64_t* oy, uint8_t* inf, size_t n, int threads) {
  return run(curve, 0, k, NULL, x, y, ox, oy, inf, n, threads);
}
/* (ox, oy) = k * G */
API int ossl_scalar_mult_base(int curve, const uint64_t* k, uint64_t* ox, uint64_t* oy, uint8_t* inf, size_t n,
                              int threads) {
  return run(curve, 1, k, NULL, NULL, NULL, ox, oy, inf, n, threads);
}
/* (ox, oy) = u1 * G + u2 * (qx, qy)   -- the ECDSA-verification shape */
API int ossl_double_scalar_mult(int curve, const uint64_t* u1, const uint64_t* u2, const uint64_t* qx,
                                const uint64_t* qy, uint64_t* ox, uint64_t* oy, uint8_t* inf, size_t n, int threads) {
  return run(curve, 2, u1, u2, qx, qy, ox, oy, inf, n, threads);
}
/* Competitor timing (benchs/p256_ref.cpp:55-91 over a batch): variable-base k*P without reading results
 * back; returns elapsed seconds, or a negative status. */
API double ossl_time_scalar_mult(int curve, const uint64_t* k, const uint64_t* x, const uint64_t* y, size_t n,
                                 int threads) {
  struct timespec a, b;
  clock_gettime(CLOCK_MONOTONIC, &a);
  int rc = run(curve, 0, k, NULL, x, y, NULL, NULL, NULL, n, threads);
  clock_gettime(CLOCK_MONOTONIC, &b);
  if (rc) return (double)rc;
  return (double)(b.tv_sec - a.tv_sec) + 1e-9 * (double)(b.tv_nsec - a.tv_nsec);
}
/* ---- ECDSA through libcrypto: the independent checker of ecsimd_hip_ecdsa_verify (tests/test_gpu_parity.py).
 * e = the digest as an integer (4 x u64 LE limbs; libcrypto receives it as 32 big-endian bytes), d = private keys. */
typedef struct {
  int curve, sign;
  const uint64_t *d, *e;
  uint64_t *r, *s, *qx, *qy;      /* sign: outputs; verify: inputs */
  uint8_t* ok;
  size_t begin, end;
  int rc;
} ecdsa_job_t;

static void le_to_be32(unsigned char* out, const uint64_t* limbs) {
  for (int i = 0; i < 32; ++i) out[i] = (unsigned char)(limbs[(31 - i) / 8] >> (8 * ((31 - i) % 8)));
}

static void* ecdsa_worker(void* arg) {
  ecdsa_job_t* j = (ecdsa_job_t*)arg;
  j->rc = -1;
  EC_GROUP* g = EC_GROUP_new_by_curve_name(nid_of(j->curve));
  BN_CTX* ctx = BN_CTX_new();
  BIGNUM *a = BN_new(), *b = BN_new();
  EC_POINT* Q = g ? EC_POINT_new(g) : NULL;
  if (!g || !ctx || !a || !b || !Q) goto done;
  for (size_t i = j->begin; i < j->end; ++i) {
    unsigned char dig[32];
    le_to_be32(dig, j->e + 4 * i);
    EC_KEY* key = EC_KEY_new();
    if (!key || !EC_KEY_set_group(key, g)) { EC_KEY_free(key); goto done; }
    if (j->sign) {
      if (!BN_lebin2bn((const unsigned char*)(j->d + 4 * i), 32, a)) { EC_KEY_free(key); goto done; }
      if (!EC_POINT_mul(g, Q, a, NULL, NULL, ctx) || !EC_KEY_set_private_key(key, a) || !EC_KEY_set_public_key(key, Q)) { EC_KEY_free(key); goto done; }
      if (!EC_POINT_get_affine_coordinates(g, Q, a, b, ctx)) { EC_KEY_free(key); goto done; }
      BN_bn2lebinpad(a, (unsigned char*)(j->qx + 4 * i), 32); BN_bn2lebinpad(b, (unsigned char*)(j->qy + 4 * i), 32);
      ECDSA_SIG* sig = ECDSA_do_sign(dig, 32, key);
      if (!sig) { EC_KEY_free(key); goto done; }
      const BIGNUM *r, *s_;
      ECDSA_SIG_get0(sig, &r, &s_);
      BN_bn2lebinpad(r, (unsigned char*)(j->r + 4 * i), 32); BN_bn2lebinpad(s_, (unsigned char*)(j->s + 4 * i), 32);
      ECDSA_SIG_free(sig);
    } else {
      int ok = 0;
      if (BN_lebin2bn((const unsigned char*)(j->qx + 4 * i), 32, a) && BN_lebin2bn((const unsigned char*)(j->qy + 4 * i), 32, b) &&
          EC_KEY_set_public_key_affine_coordinates(key, a, b)) {                   /* rejects points off the curve and coordinates >= p */
        ECDSA_SIG* sig = ECDSA_SIG_new();
        BIGNUM *r = BN_lebin2bn((const unsigned char*)(j->r + 4 * i), 32, NULL), *s_ = BN_lebin2bn((const unsigned char*)(j->s + 4 * i), 32, NULL);
        if (sig && r && s_ && ECDSA_SIG_set0(sig, r, s_)) ok = (ECDSA_do_verify(dig, 32, sig, key) == 1);     /* 0: bad signature, -1: r or s out of range */
        else { BN_free(r); BN_free(s_); }
        ECDSA_SIG_free(sig);
      }
      j->ok[i] = (uint8_t)ok;
    }
    EC_KEY_free(key);
  }
  j->rc = 0;
done:
  EC_POINT_free(Q); BN_free(a); BN_free(b); BN_CTX_free(ctx); EC_GROUP_free(g);
  return NULL;
}

static int ecdsa_run(ecdsa_job_t proto, size_t n, int threads) {
  if (nid_of(proto.curve) < 0) return -2;
  if (threads < 1) threads = 1;
  if (threads > 256) threads = 256;
  if ((size_t)threads > n) threads = n ? (int)n : 1;
  ecdsa_job_t jobs[256];
  pthread_t tid[256];
  for (int t = 0; t < threads; ++t) {
    jobs[t] = proto; jobs[t].begin = n * t / threads; jobs[t].end = n * (t + 1) / threads;
    if (threads == 1) ecdsa_worker(&jobs[t]);
    else if (pthread_create(&tid[t], NULL, ecdsa_worker, &jobs[t])) return -3;
  }
  int rc = 0;
  for (int t = 0; t < threads; ++t) { if (threads > 1) pthread_join(tid[t], NULL); if (jobs[t].rc) rc = jobs[t].rc; }
  return rc;
}
/* key pair from the private key d[i] (1 <= d < n), signature of the digest e[i]: (qx, qy) = d G, (r, s) = ECDSA_do_sign */
API int ossl_ecdsa_sign(int curve, const uint64_t* d, const uint64_t* e, uint64_t* r, uint64_t* s, uint64_t* qx, uint64_t* qy, size_t n, int threads) {
  ecdsa_job_t j = {curve, 1, d, e, r, s, qx, qy, NULL, 0, 0, 0};
  return ecdsa_run(j, n, threads);
}
/* ok[i] = ECDSA_do_verify(e[i], (r[i], s[i]), Q[i]) == 1; an invalid public key or an out-of-range r / s gives 0 */
API int ossl_ecdsa_verify(int curve, const uint64_t* e, const uint64_t* r, const uint64_t* s, const uint64_t* qx, const uint64_t* qy, uint8_t* ok, size_t n, int threads) {
  ecdsa_job_t j = {curve, 0, NULL, e, (uint64_t*)r, (uint64_t*)s, (uint64_t*)qx, (uint64_t*)qy, ok, 0, 0, 0};
  return ecdsa_run(j, n, threads);
}
API const char* ossl_version(void) { return OpenSSL_version(OPENSSL_VERSION); }
