import sys, os, faulthandler
faulthandler.enable()
sys.path.insert(0, '/root/repo/rna-mpnn_amd'); sys.path.insert(0, '/root/repo')
import torch
from rnampnn.model.rnampnn import RNAMPNN
from rnampnn.utils import synth
coords, mask, labels = synth.synth_batch([24, 17, 30, 12], first_index=70)
model = RNAMPNN(precision="f32", num_res_neighbours=8, num_res_mpnn_layers=2, padding_len=32).to("cuda:0")
model.train()
c, m, y = torch.from_numpy(coords), torch.from_numpy(mask), torch.from_numpy(labels)
onehot = torch.nn.functional.one_hot(y, 4).float()
model.manual_seed(7)
print("native", float(model.loss_and_grad(y, c, m)), flush=True)
model.manual_seed(7)
lg = model(c, m)
print("fwd ok", lg.requires_grad, float(lg.sum()), flush=True)
d = torch.ones_like(lg) * 0.01
model._train_backward_native(d)
torch.cuda.synchronize()
print("direct backward ok", float(model.flat_grad.abs().sum()), flush=True)
model.manual_seed(7)
loss = model.training_step((onehot, c, m, None))
print("loss", float(loss), flush=True)
loss.backward()
torch.cuda.synchronize()
print("autograd backward ok", float(model.flat_grad.abs().sum()), flush=True)
