function [violation,min_dist] = CheckCollEllipDMPC(p,l,n,k,E1,rmin,order)
% Shadows dmpc/matlab/CheckCollEllipDMPC.m (same signature): any neighbour closer than rmin at horizon step k (ellipsoidal metric E1).
assert(order == 2 || order == 4, 'ellipsoid order 2 or 4');   % (order 4: an all-neighbour context carries it, the dense rows are generic in it)
N = size(l,3);
prm = dmpc_params_struct(vsel(order, 5), 0.2, size(l,2), rmin, [-1 -1 0], [1 1 1], 1, 1000, 100, E1, order, -5e4);   % context only
sel = setdiff(0:N-1, n-1);
[~,~,dist] = dmpc_mex('coll_rows', prm, l, sel, k-1, k-1, p(:), [0;0;0], rmin, 1/E1(3,3), eye(3*size(l,2)));
dist = dist.^(1/(order-1));   % (the builder returns prev_dist = dist^(order-1))
violation = any(dist < rmin);
min_dist = min(dist);
end
function v = vsel(order, v2)
% the context's variant: an order-4 context is one of an all-neighbour variant (5 = solveEllipDMPC)
if order == 4, v = 5; else, v = v2; end
end
