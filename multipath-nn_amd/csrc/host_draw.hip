// Host-side input pipeline: the reference's augmentation draws (scripts/lib/data.py:24-34) replayed over raw
// MT19937 words.  No device code in this file.
//
// The reference draws, per sample and in this order on numpy's global legacy stream:
//     j    = randint(0, N)                  -- source image               (data.py:25)
//     flip = not (rand() < 0.5)             -- only if class(j) is mirror-symmetric (data.py:10-11,29)
//     du, dv = randint(-r, r + 1, 2)        -- shift                      (data.py:13-14)
// One Python call each: ~3 us per sample, 0.3-0.43 ms per batch of 128 -- the serial part of the training loop once the
// step itself takes 0.5 ms.  numpy.random.RandomState's bounded integers are a documented function of the generator's
// 32-bit outputs (numpy/random/src/distributions/distributions.c, legacy "masked rejection": draw a word, AND it with
// the smallest 2^k - 1 >= range, retry while the result exceeds the range; random_sample: (a >> 5, b >> 6) of two words ->
// (a * 2^26 + b) / 2^53).  The host therefore pulls RAW words out of numpy's OWN generator
// (randint(0, 2^32, size, uint32) returns consecutive outputs) and this function consumes them exactly as the three
// calls above would.  The stream position must come out right as well (train-adaptive-nets draws rand.choice(k_cpts)
// between two batches), and rejection sampling makes the number of words a batch needs unknowable in advance -- so the
// function is RESUMABLE and never over-draws: it is handed exactly the MINIMUM number of words the remaining draws need
// (every pending bounded integer at least one word, a pending rand() two), consumes all of them, and returns the new
// minimum (the rejections it met); the caller fetches that many more and calls again (lib/data.py:
// _draw_augmentation_fast; 640 -> 240 -> 88 -> ... words for a batch of 128, ~8 rounds).  The values drawn and the
// stream position after the batch are bit-identical to the per-call loop (tests/test_host_cpu.py; the fixtures in
// tests/golden/data_aug_golden.npz were produced by the reference module itself).
#include <cstdint>
#include "mpnn_hip.h"

static inline uint32_t mask_of(uint32_t rng) {
    uint32_t m = rng;
    m |= m >> 1; m |= m >> 2; m |= m >> 4; m |= m >> 8; m |= m >> 16;
    return m;
}

// state[0] = sample, state[1] = stage (0: j, 1: first word of rand(), 2: second word, 3: du, 4: dv),
// state[2] = first word of a rand() that is waiting for its second; state[3]: 1 = the caller vouches that EVERY entry of
// sym is set (the lower bound then counts rand()'s two words per sample: fewer rounds), else 0.  state[0..2] zero to start.
extern "C" long mpnn_draw_augmentation(const uint32_t *raw, long n_raw, int n, long n_src, const unsigned char *sym,
                                       int r_shift, int *draw, long *state) {
    if (!draw || !state || (n_raw > 0 && !raw) || n < 0 || n_src < 1 || n_src > 0xFFFFFFFFL || r_shift < 0) return MPNN_E_ARG;
    const uint32_t rng_j = (uint32_t)(n_src - 1), mask_j = mask_of(rng_j);
    const uint32_t rng_s = (uint32_t)(2 * r_shift), mask_s = mask_of(rng_s);
    const bool any_plain = sym && state[3] == 0;           // a sample may skip rand() (unless the caller knows better)
    const long per_sample_min = (rng_j ? 1 : 0) + (any_plain ? 0 : 2) + (rng_s ? 2 : 0);
    long pos = 0;
    int i = (int)state[0], stage = (int)state[1];
    while (i < n) {
        int *d = draw + 4 * i;
        if (stage == 0) {
            if (!rng_j) d[0] = 0;
            else { if (pos >= n_raw) break; const uint32_t j = raw[pos++] & mask_j; if (j > rng_j) continue; d[0] = (int)j; }
            d[1] = 0; d[2] = 0; d[3] = 0;
            stage = (!sym || sym[d[0]]) ? 1 : 3;
        } else if (stage == 1) {
            if (pos >= n_raw) break;
            state[2] = (long)raw[pos++];
            stage = 2;
        } else if (stage == 2) {
            if (pos >= n_raw) break;
            const uint32_t a = (uint32_t)state[2] >> 5, b = raw[pos++] >> 6;
            const double u = ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
            d[1] = !(u < 0.5);                             // rand_flip keeps the image when rand() < 0.5
            stage = 3;
        } else {
            if (rng_s) {
                if (pos >= n_raw) break;
                const uint32_t v = raw[pos++] & mask_s;
                if (v > rng_s) continue;
                d[stage == 3 ? 2 : 3] = (int)v - r_shift;
            }
            if (stage == 3) stage = 4; else { stage = 0; ++i; }
        }
    }
    state[0] = i; state[1] = stage;
    if (pos != n_raw) return MPNN_E_ARG;                   // handed more than the minimum: the stream would be over-drawn
    if (i >= n) return 0;
    // minimum number of words the rest of the batch needs
    long need = (long)(n - i - 1) * per_sample_min;
    if (stage == 0) need += per_sample_min > 0 ? per_sample_min : 0;
    else if (stage == 1) need += 2 + (rng_s ? 2 : 0);
    else if (stage == 2) need += 1 + (rng_s ? 2 : 0);
    else need += rng_s ? (stage == 3 ? 2 : 1) : 0;
    if (need == 0) {                                       // nothing left consumes a word (n_src == 1, r_shift == 0, no flips): finish
        return mpnn_draw_augmentation(raw, 0, n, n_src, sym, r_shift, draw, state) ;
    }
    return need;
}

// ---------------------------------------------------------------------------------------------------------------------
// The same draws from a PRIVATE MT19937 state (numpy's legacy layout: key[624] + position), generator included: no Python
// per sample or per round, and any number of batches per call -- which is what lets a net that is trained beside others
// (co-trained groups, nets sharded over ranks) see EXACTLY the batches it sees in the reference's serial experiment loop
// (scripts/train-nets:159-164: one global stream, net after net): its stream is the experiment's stream advanced over the
// draws of all iterations of the nets in front of it (draw == NULL: advance only; ~1 us per batch of 128).
// MT19937 (Matsumoto & Nishimura 1998) exactly as numpy.random.RandomState steps it: regenerate the 624 words when the
// position reaches 624, then temper key[pos++].
namespace {
struct Mt {
    uint32_t *key;
    int pos;
    inline uint32_t next() {
        if (pos >= 624) {
            for (int k = 0; k < 624; ++k) {
                const uint32_t y = (key[k] & 0x80000000u) | (key[(k + 1) % 624] & 0x7fffffffu);
                key[k] = key[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            }
            pos = 0;
        }
        uint32_t y = key[pos++];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        return y;
    }
    inline uint32_t bounded(uint32_t rng, uint32_t mask) {       // numpy's legacy masked rejection (rng > 0)
        uint32_t v;
        do v = next() & mask; while (v > rng);
        return v;
    }
};
}  // namespace

extern "C" long mpnn_draw_augmentation_mt(uint32_t *key, int *pos, long batches, int n, long n_src, const unsigned char *sym,
                                          int r_shift, int *draw) {
    if (!key || !pos || *pos < 0 || *pos > 624 || batches < 0 || n < 0 || n_src < 1 || n_src > 0xFFFFFFFFL || r_shift < 0) return MPNN_E_ARG;
    Mt g{key, *pos};
    const uint32_t rng_j = (uint32_t)(n_src - 1), mask_j = mask_of(rng_j);
    const uint32_t rng_s = (uint32_t)(2 * r_shift), mask_s = mask_of(rng_s);
    for (long b = 0; b < batches; ++b)
        for (int i = 0; i < n; ++i) {
            const uint32_t j = rng_j ? g.bounded(rng_j, mask_j) : 0u;
            int flip = 0;
            if (!sym || sym[j]) {
                const uint32_t a = g.next() >> 5, c = g.next() >> 6;
                const double u = ((double)a * 67108864.0 + (double)c) / 9007199254740992.0;
                flip = !(u < 0.5);
            }
            const int du = rng_s ? (int)g.bounded(rng_s, mask_s) - r_shift : 0;
            const int dv = rng_s ? (int)g.bounded(rng_s, mask_s) - r_shift : 0;
            if (draw) {
                int *d = draw + ((size_t)b * n + i) * 4;
                d[0] = (int)j; d[1] = flip; d[2] = du; d[3] = dv;
            }
        }
    *pos = g.pos;
    return 0;
}
