"""Host logic of mlp.claim_grad_slot (no GPU): the persistent gradient-bucket segment of an engine is handed out at most
once per autograd pass and never while a parameter's .grad still lives in it."""
import torch

from neuraludf_amd import mlp


class _Layer:
    def __init__(self, p):
        self.p = p

    def params(self):
        return [self.p]


class _Engine:
    pass


def _setup():
    w = torch.nn.Parameter(torch.ones(4))
    eng = _Engine()
    eng.grad_slot = torch.zeros(4)
    log = []

    class F(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, w):
            return x * w

        @staticmethod
        def backward(ctx, g):
            slot = mlp.claim_grad_slot(eng, [_Layer(w)])
            log.append(slot is not None)
            out = slot[0:4].view(4) if slot is not None else torch.empty(4)     # a fresh view, as unpack_group makes
            out.copy_(g)
            return None, out
    return w, eng, log, F


def test_slot_is_claimed_once_per_autograd_pass():
    w, eng, log, F = _setup()
    x = torch.ones(4)
    (F.apply(x, w).sum() * 2 + F.apply(x, w).sum() * 3).backward()        # two evaluations in one graph
    assert sorted(log) == [False, True]
    assert torch.equal(w.grad, torch.full((4,), 5.0))                      # 2 + 3, not 2 * 3
    assert eng._slot_inflight is False                                     # released when the pass ended


def test_slot_is_not_reused_while_a_grad_lives_in_it():
    w, eng, log, F = _setup()
    x = torch.ones(4)
    (F.apply(x, w).sum() * 2).backward()
    assert log == [True] and w.grad.data_ptr() == eng.grad_slot.data_ptr()  # p.grad IS the bucket view
    (F.apply(x, w).sum() * 3).backward()                                    # accumulation: must not overwrite the view
    assert log == [True, False]
    assert torch.equal(w.grad, torch.full((4,), 5.0))
    w.grad.zero_()                                                          # zero_grad(set_to_none=False)
    (F.apply(x, w).sum() * 7).backward()
    assert log[-1] is False and torch.equal(w.grad, torch.full((4,), 7.0))
    w.grad = None                                                           # zero_grad(set_to_none=True): fast path back
    (F.apply(x, w).sum() * 4).backward()
    assert log[-1] is True and torch.equal(w.grad, torch.full((4,), 4.0))
    assert w.grad.data_ptr() == eng.grad_slot.data_ptr()
