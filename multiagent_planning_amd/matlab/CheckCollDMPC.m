function violation = CheckCollDMPC(p,l,n,k,r_min)
% Shadows dmpc/matlab/CheckCollDMPC.m (same signature): any other agent closer than r_min (Euclidean) at horizon step k.
N = size(l,3);
sel = setdiff(0:N-1, n-1);
violation = false;
if isempty(sel), return; end
prm = dmpc_params_struct(0, 0.2, size(l,2), r_min, [-1 -1 0], [1 1 1], 1, 1000, 100, eye(3), 2, -5e4);   % context only
[~,~,dist] = dmpc_mex('coll_rows', prm, l, sel, k-1, k-1, p(:), [0;0;0], r_min, 1, eye(3*size(l,2)));
violation = any(dist < r_min);
end
