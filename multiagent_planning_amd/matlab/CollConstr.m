function [Ain_total, bin_total] = CollConstr(p,po,k,l,Ain,rmin,E1,E2,order)
% Shadows dec-iSCP/CollConstr.m (same signature): rows of time step k against every obstacle in l, built on the GPU.
assert(order == 2 || order == 4, 'ellipsoid order 2 or 4');   % (order 4: an all-neighbour context carries it, the dense rows are generic in it)
if isempty(l)
    Ain_total = zeros(0,size(Ain,1)); bin_total = zeros(0,1); return
end
prm = dmpc_params_struct(vsel(order, 0), 0.2, 15, rmin, [0 0 0], [0 0 0], 1, 1000, 100, E1, order, -5e4);   % context only
[Ain_total, bin_total] = dmpc_mex('coll_rows', prm, l, 0:size(l,3)-1, k-1, k-2, p(:), po(:), rmin, 1/E1(3,3), Ain);
end
function v = vsel(order, v2)
% the context's variant: an order-4 context is one of an all-neighbour variant (5 = solveEllipDMPC)
if order == 4, v = 5; else, v = v2; end
end
