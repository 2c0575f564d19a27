function tol = maxDeviation(p, prev_p)
% Shadows dmpc/matlab/maxDeviation.m (same signature) over dmpc_mex; as the reference, only the first length(p)/3 columns are looked at.
prm = dmpc_params_struct(0, 0.2, 15, 0.35, [-1 -1 0], [1 1 1], 1, 1000, 100, eye(3), 2, -5e4);   % context only
tol = dmpc_mex('max_deviation', prm, p, prev_p);
end
