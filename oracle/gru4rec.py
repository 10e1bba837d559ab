"""CPU restatement of the reference's GRU4RecPlus training / inference step -- TEST INFRASTRUCTURE ONLY
(tests/ and __graft_entry__.smoke() may import it; the product never does).

PARITY UNPINNED.  The reference implements this model on TensorFlow 1.14 (`README.md:97`), which is not
installed here and whose source is not under /root/reference; the reference holds no test, fixture or
recorded output for it.  This file therefore restates

* the graph of `skrec/recommender/GRU4RecPlus.py:124-200` (variables :124-135, `_softmax_neg` :137-144,
  `_bpr_max_loss` :146-156, `_top1_max_loss` :158-166, `_build_model` :168-196) and
* the published semantics of the two TF-1.14 pieces it calls:
  `tf.nn.rnn_cell.GRUCell.call` (gate kernel [in+h, 2h] with bias initialised to 1.0, split into r | u;
  candidate kernel [in+h, h] applied to [x, r*h]; new_h = u*h + (1-u)*c; glorot-uniform kernels) and
  `tf.train.AdamOptimizer` (lr_t = lr*sqrt(1-b2^t)/(1-b1^t); p -= lr_t*m/(sqrt(v)+eps); embedding
  gradients arrive as IndexedSlices but `_apply_sparse` still decays m, v and moves EVERY row)

in float32 torch on the CPU, with autograd providing the gradients the HIP kernels derive by hand.
The recurrent state enters the graph through placeholders (`state_ph`, :127), so no gradient flows into
earlier time steps: every step is a one-step truncated BPTT.
"""
import numpy as np
import torch

ACTS = {"tanh": torch.tanh, "relu": torch.relu}


def final_act(x, kind):
    if kind == "linear":
        return x
    if kind == "relu":
        return torch.relu(x)
    if kind == "leaky_relu":
        return torch.nn.functional.leaky_relu(x, 0.2)       # tf.nn.leaky_relu default alpha
    raise ValueError(kind)


def gru_cell(x, h, Wg, bg, Wc, bc, act):
    """tf.nn.rnn_cell.GRUCell.call (TF 1.14)"""
    H = h.shape[1]
    gates = torch.sigmoid(torch.cat([x, h], 1) @ Wg + bg)
    r, u = gates[:, :H], gates[:, H:]
    c = ACTS[act](torch.cat([x, r * h], 1) @ Wc + bc)
    return u * h + (1.0 - u) * c


def softmax_neg(logits):
    """GRU4RecPlus.py:137-144"""
    b, n = logits.shape
    hm = 1.0 - torch.eye(b, n, dtype=logits.dtype)
    lg = logits * hm
    lg = lg - lg.max(dim=1, keepdim=True).values
    e = torch.exp(lg) * hm
    return e / e.sum(dim=1, keepdim=True)


def bpr_max_loss(logits, bpr_reg):
    """GRU4RecPlus.py:146-156"""
    s = softmax_neg(logits)
    pos = torch.diagonal(logits).reshape(-1, 1)
    prob = (torch.sigmoid(pos - logits) * s).sum(1)
    loss = -torch.log(prob + 1e-24)
    reg = (logits.pow(2) * s).sum(1)
    return (loss + bpr_reg * reg).mean()


def top1_max_loss(logits):
    """GRU4RecPlus.py:158-166"""
    s = softmax_neg(logits)
    pos = torch.diagonal(logits).reshape(-1, 1)
    prob = torch.sigmoid(-pos + logits) + torch.sigmoid(logits.pow(2))
    return (prob * s).sum(1).mean()


class GRU4RecOracle(object):
    """parameters are float32 CPU tensors; `cells` = [(Wg, bg, Wc, bc), ...] bottom layer first"""

    def __init__(self, E_in, cells, E_out, b_out, hidden_act="tanh", final="linear", loss="bpr_max", bpr_reg=1.0,
                 reg=0.0, lr=1e-3):
        t = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32).clone().requires_grad_(True)  # noqa: E731
        self.E_in, self.E_out, self.b_out = t(E_in), t(E_out), t(b_out)
        self.cells = [tuple(t(w) for w in cell) for cell in cells]
        self.hidden_act, self.final, self.loss_kind = hidden_act, final, loss
        self.bpr_reg, self.reg, self.lr = float(bpr_reg), float(reg), float(lr)
        self.params = [self.E_in] + [w for cell in self.cells for w in cell] + [self.E_out, self.b_out]
        self.m = [torch.zeros_like(p) for p in self.params]
        self.v = [torch.zeros_like(p) for p in self.params]
        self.t = 0

    def forward(self, X, states):
        """-> (output of the top layer, new states); X int64 [b], states list of [b, h_l]"""
        x = self.E_in[torch.as_tensor(X, dtype=torch.long)]
        inputs = x
        new_states = []
        for (Wg, bg, Wc, bc), h in zip(self.cells, states):
            x = gru_cell(x, torch.as_tensor(h, dtype=torch.float32), Wg, bg, Wc, bc, self.hidden_act)
            new_states.append(x)
        return x, new_states, inputs

    def loss(self, X, Y, states):
        out, new_states, inputs = self.forward(X, states)
        Y = torch.as_tensor(Y, dtype=torch.long)
        items, bias = self.E_out[Y], self.b_out[Y]
        logits = final_act(out @ items.t() + bias, self.final)
        main = bpr_max_loss(logits, self.bpr_reg) if self.loss_kind == "bpr_max" else top1_max_loss(logits)
        l2 = 0.5 * (inputs.pow(2).sum() + items.pow(2).sum() + bias.pow(2).sum())     # utils/tf1x.py:25-29
        return main + self.reg * l2, main, new_states, logits

    def grads(self, X, Y, states):
        for p in self.params:
            p.grad = None
        total, main, new_states, logits = self.loss(X, Y, states)
        total.backward()
        g = [p.grad if p.grad is not None else torch.zeros_like(p) for p in self.params]
        return float(main.detach()), [s.detach() for s in new_states], [x.detach().clone() for x in g], logits.detach()

    def train_step(self, X, Y, states, b1=0.9, b2=0.999, eps=1e-8):
        """one `sess.run([update_opt, final_state])` (GRU4RecPlus.py:231): returns (loss, new states)"""
        main, new_states, g, _ = self.grads(X, Y, states)
        self.t += 1
        lr_t = self.lr * np.sqrt(1.0 - b2 ** self.t) / (1.0 - b1 ** self.t)
        with torch.no_grad():
            for p, gi, m, v in zip(self.params, g, self.m, self.v):
                m.mul_(b1).add_(gi, alpha=1.0 - b1)
                v.mul_(b2).addcmul_(gi, gi, value=1.0 - b2)
                p.sub_(np.float32(lr_t) * m / (v.sqrt() + np.float32(eps)))
        return main, new_states

    def user_embeddings(self, rowptr, items_by_time):
        """_get_user_embeddings (GRU4RecPlus.py:256-302): every user's history through the stack from a zero
        state; the result does not depend on how the reference batches the users"""
        n_users = len(rowptr) - 1
        out = np.zeros((n_users, self.cells[-1][2].shape[1]), np.float32)
        with torch.no_grad():
            for u in range(n_users):
                seq = items_by_time[rowptr[u]:rowptr[u + 1]]
                if len(seq) == 0:
                    continue
                states = [torch.zeros(1, cell[2].shape[1]) for cell in self.cells]
                for it in seq:
                    o, states, _ = self.forward(np.array([it]), states)
                out[u] = o[0].numpy()
        return out
