import sys, os
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, ray_tracing_amd as rta, scenes
ctx=rta.Context(0)
g0,o=scenes.build_icosahedron(leaf_refine=None)
rays,w,h=o.camera_rays(scenes.oracle_camera(scenes.C1_CAMERA))
r=rays[[242340,242980]]
tr=g0.triangles()
for k in (929,930):
    print('tri prim',k+2, tr[k]['v'].tolist())
for refine in (None, 8, 16, 64, 0xFFFFFFFF):
    g,_=scenes.build_icosahedron(leaf_refine=refine, oracle=False)
    hits=g.first_inter_pt(ctx,r)
    hb=g.first_inter_pt(ctx,r,use_tree=False)
    print(refine, hits['prim_id'], hits['t'], 'brute', hb['prim_id'])
import oracle_bind as ob
for k in (931,932):
    pt=np.zeros(3); uv=np.zeros(2)
    ob.L.orc_object_first_inter_pt(o.h,k,r[0].ctypes.data,pt.ctypes.data,uv.ctypes.data)
    print('oracle prim',k,pt, np.linalg.norm(pt-r[0,:3]))
