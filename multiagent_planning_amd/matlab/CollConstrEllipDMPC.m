function [Ain_total, bin_total, prev_dist] = CollConstrEllipDMPC(p,po,vo,n,k,l,rmin,Ain,A_initp,E1,E2,order)
% Shadows dmpc/matlab/CollConstrEllipDMPC.m (same signature): one row per neighbour j ~= n, evaluated and constraining at step k.
assert(order == 2 || order == 4, 'ellipsoid order 2 or 4');   % (order 4: an all-neighbour context carries it, the dense rows are generic in it)
N = size(l,3);
sel = setdiff(0:N-1, n-1);
a0 = A_initp(3*(k-1)+1:3*k,:)*[po(:); vo(:)];
prm = dmpc_params_struct(vsel(order, 5), 0.2, size(l,2), rmin, [-1 -1 0], [1 1 1], 1, 1000, 100, E1, order, -5e4);   % context only
[Ain_total, bin_total, prev_dist] = dmpc_mex('coll_rows', prm, l, sel, k-1, k-1, p(:), a0, rmin, 1/E1(3,3), Ain);
end
function v = vsel(order, v2)
% the context's variant: an order-4 context is one of an all-neighbour variant (5 = solveEllipDMPC)
if order == 4, v = 5; else, v = v2; end
end
