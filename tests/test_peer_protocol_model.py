"""A model of the direct peer exchange's hand-shake (srl_kernels.hip peer_exchange), run on the CPU: G ranks, each with
an inbox of 2 slots x G rows of tagged words.  Exchange e: a rank stores its row, tagged e, into slot e & 1 of EVERY inbox (its own
included), then polls its own inbox until the rows of all ranks carry tag e, adds them in rank order and moves on.  There is no
barrier, no acknowledgement and nothing is ever reset -- what makes two slots enough is that a rank can only be one exchange
ahead of any peer (it needs that peer's row of exchange e + 1, which the peer stores only after finishing exchange e).
The model checks exactly that claim under random scheduling: every rank gets the right sum in every exchange, nobody reads a row
of the wrong exchange, nobody waits forever -- and that ONE slot would not be enough (the model catches the overwrite)."""
import random
import threading
import time

import pytest


class Inbox:
    def __init__(self, G, slots):
        self.tag = [[0] * G for _ in range(slots)]          # tag 0 = "nothing here": exchange numbers start at 1
        self.val = [[0.0] * G for _ in range(slots)]


def _run(G, exchanges, slots, seed, jitter=2e-4):
    rng = random.Random(seed)
    inbox = [Inbox(G, slots) for _ in range(G)]
    delays = [[rng.random() * jitter * (1 + 4 * (rng.random() < 0.1)) for _ in range(exchanges + 1)] for _ in range(G)]
    results = [[None] * (exchanges + 1) for _ in range(G)]
    errors = []

    def rank(r):
        try:
            for e in range(1, exchanges + 1):
                time.sleep(delays[r][e])                      # the rank's kernel of this pass
                mine = float(r + 1) * e
                s = e % slots
                for peer in range(G):                         # store the row into every inbox: value first, tag last (the tag IS the flag;
                    inbox[peer].val[s][r] = mine              # on the device value and tag travel in one 8-byte granule)
                    inbox[peer].tag[s][r] = e
                total, t0 = 0.0, time.time()
                for src in range(G):                          # collect in rank order
                    while inbox[r].tag[s][src] != e:
                        if inbox[r].tag[s][src] > e:
                            raise AssertionError(f"rank {r}: row of rank {src} already carries exchange {inbox[r].tag[s][src]} while waiting for {e}")
                        if time.time() - t0 > 20:
                            raise TimeoutError(f"rank {r} starved in exchange {e} waiting for rank {src}")
                        time.sleep(0)
                    total += inbox[r].val[s][src]
                results[r][e] = total
        except Exception as ex:  # noqa: BLE001
            errors.append(ex)

    ts = [threading.Thread(target=rank, args=(r,)) for r in range(G)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    return results, errors


@pytest.mark.parametrize("G", [2, 4, 8])
def test_two_slots_carry_any_schedule(G):
    exchanges = 150
    results, errors = _run(G, exchanges, slots=2, seed=100 + G)
    assert not errors, errors[:2]
    for e in range(1, exchanges + 1):
        want = sum(float(r + 1) * e for r in range(G))
        assert all(results[r][e] == want for r in range(G)), e     # the same sum, the same bits, on every rank


def test_one_slot_is_not_enough():
    """With a single slot a fast rank overwrites its row of exchange e with the one of e + 1 before a slow peer has read it: the model
    must notice (a newer tag than the one waited for), otherwise it would prove nothing about the two-slot form."""
    seen = False
    for seed in range(40):
        _, errors = _run(4, 60, slots=1, seed=seed, jitter=3e-4)
        if any(isinstance(x, AssertionError) for x in errors):
            seen = True
            break
    assert seen
