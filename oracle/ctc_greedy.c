/* ORACLE (test infrastructure, never shipped): C restatement of the reference's CTC greedy decode.
 *   argmax per frame      : src/ctc/ctc.py:180-188   (torch.argmax -> lowest index on ties)
 *   collapse + drop blank : src/models/maskctc_model.py:289-291 (itertools.groupby, y_hat != 0)
 * Integer path: the HIP kernel must match this bit for bit.  Built by __graft_entry__.build() into
 * oracle/_build/liboracle_ctc.so and used only by tests/ and smoke().
 */
#include <stdint.h>

/* logits[t*V + v], t < T ; ids[T] out ; hyp[T] out ; returns hypothesis length */
int64_t oracle_ctc_greedy(const float* logits, int64_t T, int64_t V, int64_t blank, int64_t* ids, int64_t* hyp) {
  int64_t n = 0, prev = -1;
  for (int64_t t = 0; t < T; ++t) {
    const float* row = logits + t * V;
    int64_t best = 0;
    for (int64_t v = 1; v < V; ++v)
      if (row[v] > row[best]) best = v; /* strict > keeps the lowest index on ties */
    ids[t] = best;
    if (best != prev && best != blank) hyp[n++] = best;
    prev = best;
  }
  return n;
}
