import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, torch
import golden_util as gu
from oracle import agcn_oracle as orc
from model.agcn import Model
from agcn_amd import ops
dev = torch.device('cuda:0')
name = 'm_kin_b2'
gold = gu.load(name)
n, v, num_class, seed, t = [int(i) for i in gold['meta']]
model = Model(num_class=num_class, num_point=v, num_person=2, graph='graph.kinetics.Graph', graph_args=dict(labeling_mode='spatial'))
sd = orc.randomized_state(orc.model_param_shapes(num_class, v), seed, stress=float(gold['meta.stress']))
model.load_state_dict(sd); model.to(dev).eval()
xn, lab = gu.model_inputs(n, v, num_class, seed, t)
x = torch.from_numpy(xn).to(dev)
mx = {}
orig = ops.conv9_infer
def spy(xx, w, b, res=None, relu=True, stride=1):
    mx.setdefault('x', []).append(float(xx.abs().max())); mx.setdefault('w', []).append(float(w.abs().max()))
    return orig(xx, w, b, res, relu, stride)
ops.conv9_infer = spy
with torch.no_grad():
    le = model(x)
print('stress', float(gold['meta.stress']), 'err', gu.rel_err(le.cpu().numpy(), gold['logits_eval']))
print('max |x| per conv', ['%.3g' % a for a in mx['x']])
print('max |w| per conv', ['%.3g' % a for a in mx['w']])
