import os, sys, pickle
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import zkevm_circuits_amd as z
from zkevm_circuits_amd import plonk
from oracle import cref
from plonk_fixtures import build_circuit
ctx = z.Context(0)
k, wide, seed = int(sys.argv[1]), sys.argv[2] == '1', int(sys.argv[3])
S = 0x5EC2E7
circ, adv, inst = build_circuit(k, seed, wide)
srs = ctx.srs_setup_with_s(k, cref.fr_const(S))
pk = ctx.pk_create(srs, circ.blob())
com, rep = pk.vk(circ.F + len(circ.perm_cols))
proof = ctx.create_proof(pk, [plonk.column_to_mont(c) for c in adv], [plonk.column_to_mont(c) for c in inst], bytes(16))
os.makedirs('gpurun_out', exist_ok=True)
pickle.dump({'proof': proof, 'vk': cref.affine_from_mont(com), 'vk_repr': cref.from_mont(rep.reshape(1, 4))[0], 'args': (k, wide, seed)}, open('gpurun_out/proof_dump.pkl', 'wb'))
print('proof bytes', len(proof))
