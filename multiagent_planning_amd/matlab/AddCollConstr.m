function [Ain_total, bin_total] = AddCollConstr(p,po,K,rmin,A,E1,E2,order)
% Shadows cup-SCP/AddCollConstr.m (same signature): the K*N*(N-1)/2 pairwise collision rows, built on the GPU.
assert(order == 2 || order == 4, 'ellipsoid order 2 or 4');   % (order 4: an all-neighbour context carries it, the dense rows are generic in it)
prm = dmpc_params_struct(vsel(order, 0), 0.2, 15, rmin, [0 0 0], [0 0 0], 1, 1000, 100, E1, order, -5e4);   % context only
[Ain_total, bin_total] = dmpc_mex('add_coll_constr', prm, p, po, rmin, 1/E1(3,3), A);
end
function v = vsel(order, v2)
% the context's variant: an order-4 context is one of an all-neighbour variant (5 = solveEllipDMPC)
if order == 4, v = 5; else, v = v2; end
end
