function [Ain_total, bin_total] = AddCollConstr(p,po,K,rmin,A,E1,E2,order)
% Shadows cup-SCP/AddCollConstr.m (same signature): the K*N*(N-1)/2 pairwise collision rows, built on the GPU.
assert(order == 2, 'only order = 2 is supported');
prm = dmpc_params_struct(0, 0.2, 15, rmin, [0 0 0], [0 0 0], 1, 1000, 100, E1, order, -5e4);   % context only
[Ain_total, bin_total] = dmpc_mex('add_coll_constr', prm, p, po, rmin, 1/E1(3,3), A);
end
